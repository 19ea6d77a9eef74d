// host_math.hpp -- the small amount of dense algebra the HOST side of the product needs
// (map bookkeeping between Matches; nothing here is on the per-point hot path):
// (the 3x3 routines are __host__ __device__: the device NDT map update, kernels_ndt_update.hpp, runs the same code)
//   * keyframe gate IsNeedAddCloud: R_last^-1 * R, RPY            icp_optimized.h:218-234
//   * IncrementalNDT::UpdateVoxel: mean/cov, (S + eps I)^-1, SVD  incremental_ndt.h:91-179
//   * float cloud transform for map updates                       pointcloud_utility.h:141-195
#pragma once
#include "host_maps.hpp"

namespace fls {
namespace hm {

__host__ __device__ inline void mul3(const double* A, const double* B, double* C) {  // column-major 3x3
    double T[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) T[i + j * 3] = (A[i] * B[j * 3] + A[i + 3] * B[1 + j * 3]) + A[i + 6] * B[2 + j * 3];
    for (int k = 0; k < 9; ++k) C[k] = T[k];
}
__host__ __device__ inline double cof3(const double* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 + j1 * 3] * m[i2 + j2 * 3] - m[i1 + j2 * 3] * m[i2 + j1 * 3];
}
__host__ __device__ inline void inv3(const double* m, double* inv) {  // closed-form cofactor inverse (Eigen Matrix3d::inverse)
    const double c00 = cof3(m, 0, 0), c10 = cof3(m, 1, 0), c20 = cof3(m, 2, 0);
    const double invdet = 1.0 / ((c00 * m[0] + c10 * m[1]) + c20 * m[2]);
    inv[0] = c00 * invdet; inv[3] = c10 * invdet; inv[6] = c20 * invdet;
    inv[1] = cof3(m, 0, 1) * invdet; inv[4] = cof3(m, 1, 1) * invdet; inv[7] = cof3(m, 2, 1) * invdet;
    inv[2] = cof3(m, 0, 2) * invdet; inv[5] = cof3(m, 1, 2) * invdet; inv[8] = cof3(m, 2, 2) * invdet;
}
__host__ __device__ inline double nrm3(const double* v) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

// two-sided Jacobi SVD of a 3x3, full U and V, singular values descending
__host__ __device__ inline void svd3(const double* A, double* U, double* S, double* V) {
    const double tiny = std::numeric_limits<double>::min(), prec = 2.0 * std::numeric_limits<double>::epsilon();
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(A[i]));
    if (scale == 0.0) scale = 1.0;
    double W[9];
    for (int i = 0; i < 9; ++i) { W[i] = A[i] / scale; U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    double maxd = std::max(std::fabs(W[0]), std::max(std::fabs(W[4]), std::fabs(W[8])));
    bool done = false;
    for (int sweep = 0; sweep < 64 && !done; ++sweep) {
        done = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double thr = std::max(tiny, prec * maxd);
                if (!(std::fabs(W[p + q * 3]) > thr || std::fabs(W[q + p * 3]) > thr)) continue;
                done = false;
                const double m00 = W[p + p * 3], m01 = W[p + q * 3], m10 = W[q + p * 3], m11 = W[q + q * 3];
                double c1, s1;
                const double t = m00 + m11, d = m10 - m01;
                if (std::fabs(d) < tiny) { s1 = 0.0; c1 = 1.0; }
                else { const double u = t / d, tmp = std::sqrt(1.0 + u * u); s1 = 1.0 / tmp; c1 = u / tmp; }
                const double n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11, n11 = -s1 * m01 + c1 * m11;
                double jc, js;
                const double deno = 2.0 * std::fabs(n01);
                if (deno < tiny) { jc = 1.0; js = 0.0; }
                else {
                    const double tau = (n00 - n11) / deno, w = std::sqrt(tau * tau + 1.0);
                    const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
                    const double sg = tt > 0.0 ? 1.0 : -1.0, nn = 1.0 / std::sqrt(tt * tt + 1.0);
                    js = -sg * (n01 / std::fabs(n01)) * std::fabs(tt) * nn;
                    jc = nn;
                }
                const double lc = c1 * jc + s1 * js, ls = s1 * jc - c1 * js;  // rot1 * j_right^T
                for (int k = 0; k < 3; ++k) {
                    const double xi = W[p + k * 3], yi = W[q + k * 3];
                    W[p + k * 3] = lc * xi + ls * yi;
                    W[q + k * 3] = -ls * xi + lc * yi;
                    const double ux = U[k + p * 3], uy = U[k + q * 3];
                    U[k + p * 3] = lc * ux + ls * uy;
                    U[k + q * 3] = -ls * ux + lc * uy;
                }
                for (int k = 0; k < 3; ++k) {
                    const double xi = W[k + p * 3], yi = W[k + q * 3];
                    W[k + p * 3] = jc * xi - js * yi;
                    W[k + q * 3] = js * xi + jc * yi;
                    const double vx = V[k + p * 3], vy = V[k + q * 3];
                    V[k + p * 3] = jc * vx - js * vy;
                    V[k + q * 3] = js * vx + jc * vy;
                }
                maxd = std::max(maxd, std::max(std::fabs(W[p + p * 3]), std::fabs(W[q + q * 3])));
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W[i + i * 3];
        S[i] = std::fabs(a) * scale;
        if (a < 0.0) for (int k = 0; k < 3; ++k) U[k + i * 3] = -U[k + i * 3];
    }
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int k = i + 1; k < 3; ++k) if (S[k] > S[pos]) pos = k;
        if (S[pos] == 0.0) break;
        if (pos != i) {
            { const double t = S[i]; S[i] = S[pos]; S[pos] = t; }
            for (int k = 0; k < 3; ++k) {
                const double tu = U[k + i * 3]; U[k + i * 3] = U[k + pos * 3]; U[k + pos * 3] = tu;
                const double tv = V[k + i * 3]; V[k + i * 3] = V[k + pos * 3]; V[k + pos * 3] = tv;
            }
        }
    }
}

// Unpivoted LDL^T solve of a symmetric positive definite 6x6 system H x = g (column-major H), static indices only: the fast path
// of the Gauss-Newton tails (kernels_p2plane.hpp).  false when a pivot is not safely positive (d_min <= 1e-9 d_max, or NaN):
// the caller then runs the restated Eigen solver, whose rank-revealing behaviour is the reference's semantics for such systems.
__host__ __device__ inline bool ldlt_solve6(const double* H, const double* g, double* x) {
    double L[6][6], D[6], y[6];
    double dmax = 0.0, dmin = 1.0e300;
    bool positive = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = H[j + 6 * j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= (L[j][k] * L[j][k]) * D[k];
        D[j] = d;
        positive = positive && (d > 0.0);  // false for NaN as well
        dmax = d > dmax ? d : dmax;
        dmin = d < dmin ? d : dmin;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = H[i + 6 * j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= (L[i][k] * L[j][k]) * D[k];
            L[i][j] = s * inv;
        }
    }
    if (!positive || !(dmin > 1.0e-9 * dmax)) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = y[i] / D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * y[k];
        y[i] = s;
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) { x[i] = y[i]; finite = finite && (y[i] - y[i] == 0.0); }
    return finite;  // a non-finite right-hand side goes through the exact solver as well
}

// ---- IncrementalNDT voxel statistics (incremental_ndt.h:91-179), shared by the host path (matcher_ndt.hpp) and the device
// map update (kernels_ndt_update.hpp): the same code, the same IEEE operation sequence on both sides ----
template <class GetPt>
__host__ __device__ inline void ndt_mean_cov(const int len, GetPt get, double* mean, double* cov) {  // ComputeMeanAndCov :91-110
    double s[3] = {0.0, 0.0, 0.0}, q[3];
    for (int k = 0; k < len; ++k) { get(k, q); s[0] += q[0]; s[1] += q[1]; s[2] += q[2]; }
    for (int a = 0; a < 3; ++a) mean[a] = s[a] / double(len);
    double c[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < len; ++k) {
        get(k, q);
        const double v[3] = {q[0] - mean[0], q[1] - mean[1], q[2] - mean[2]};
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) c[i + j * 3] += v[i] * v[j];
    }
    for (int k = 0; k < 9; ++k) cov[k] = c[k] / double(len - 1);
}
__host__ __device__ inline void ndt_regularised_info(const double* sigma, double* info) {  // (sigma + 1e-3 I)^-1 :144, :155
    double m[9];
    for (int k = 0; k < 9; ++k) m[k] = sigma[k] + ((k % 4 == 0) ? 1.0 : 0.0) * 1.0e-3;
    inv3(m, info);
}
// UpdateMeanAndCov :112-120 followed by the SVD clamp and info = V diag(1/s) U^T :160-176
__host__ __device__ inline void ndt_merge(double* mu, double* sigma, double* info, const int hist_n, const double* cmu, const double* cvar, const int cn) {
    double nmu[3], nvar[9];
    for (int a = 0; a < 3; ++a) nmu[a] = (double(hist_n) * mu[a] + double(cn) * cmu[a]) / double(hist_n + cn);
    const double dh[3] = {mu[0] - nmu[0], mu[1] - nmu[1], mu[2] - nmu[2]};
    const double dc[3] = {cmu[0] - nmu[0], cmu[1] - nmu[1], cmu[2] - nmu[2]};
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            nvar[i + j * 3] = (double(hist_n) * (sigma[i + j * 3] + dh[i] * dh[j]) + double(cn) * (cvar[i + j * 3] + dc[i] * dc[j])) / double(hist_n + cn);
    for (int a = 0; a < 3; ++a) mu[a] = nmu[a];
    for (int k = 0; k < 9; ++k) sigma[k] = nvar[k];
    double U[9], S[3], V[9];
    svd3(sigma, U, S, V);
    if (S[1] < S[0] * 1e-3) S[1] = S[0] * 1e-3;
    if (S[2] < S[0] * 1e-3) S[2] = S[0] * 1e-3;
    const double il[3] = {1.0 / S[0], 1.0 / S[1], 1.0 / S[2]};
    double VL[9];
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) VL[i + k * 3] = V[i + k * 3] * il[k];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) info[i + j * 3] = (VL[i] * U[j] + VL[i + 3] * U[j + 3]) + VL[i + 6] * U[j + 6];
}

// TransformPointCloud(cloud, Mat4d): R,t -> float, then float r0*x + (r1*y + r2*z) + t
inline std::vector<PtI> xform_cloud_f(const std::vector<PtI>& c, const double* T) {
    float R[9], t[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + j * 3] = float(T[i + j * 4]);
    for (int i = 0; i < 3; ++i) t[i] = float(T[12 + i]);
    std::vector<PtI> o(c.size());
    for (size_t k = 0; k < c.size(); ++k) {
        const PtI& p = c[k];
        o[k].x = (R[0] * p.x + (R[3] * p.y + R[6] * p.z)) + t[0];
        o[k].y = (R[1] * p.x + (R[4] * p.y + R[7] * p.z)) + t[1];
        o[k].z = (R[2] * p.x + (R[5] * p.y + R[8] * p.z)) + t[2];
        o[k].i = p.i;
    }
    return o;
}
// pcl::transformPointCloud(cloud, out, Matrix4d): double evaluation, float result
inline std::vector<PtI> xform_cloud_d(const std::vector<PtI>& c, const double* T) {
    std::vector<PtI> o(c.size());
    for (size_t k = 0; k < c.size(); ++k) {
        const double x = c[k].x, y = c[k].y, z = c[k].z;
        o[k].x = float(((T[0] * x + T[4] * y) + T[8] * z) + T[12]);
        o[k].y = float(((T[1] * x + T[5] * y) + T[9] * z) + T[13]);
        o[k].z = float(((T[2] * x + T[6] * y) + T[10] * z) + T[14]);
        o[k].i = c[k].i;
    }
    return o;
}

// IsNeedAddCloud; the reference's function-static last_T is per handle
struct KeyframeGate {
    bool init = false;
    double last_T[16];
    bool need(const double* T, double dist_thr, double rot_thr) {
        if (!init) { std::memcpy(last_T, T, sizeof(last_T)); init = true; }
        double Rl[9], Rc[9], Rli[9], Rd[9];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) { Rl[i + j * 3] = last_T[i + j * 4]; Rc[i + j * 3] = T[i + j * 4]; }
        inv3(Rl, Rli);
        mul3(Rli, Rc, Rd);
        const double roll = std::atan2(Rd[2 + 1 * 3], Rd[2 + 2 * 3]), pitch = std::asin(-Rd[2]), yaw = std::atan2(Rd[1], Rd[0]);
        const double dt[3] = {T[12] - last_T[12], T[13] - last_T[13], T[14] - last_T[14]};
        if (nrm3(dt) > dist_thr || std::fabs(roll) > rot_thr || std::fabs(pitch) > rot_thr || std::fabs(yaw) > rot_thr) {
            std::memcpy(last_T, T, sizeof(last_T));
            return true;
        }
        return false;
    }
};

}  // namespace hm
}  // namespace fls
