// ============================================================================
// oracle/flo_loop.h  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// The loop-closure matcher, LoopClosure::Match (src/slam/loop_closure.cpp:233-267): four-resolution
// pcl::NormalDistributionsTransform (10 / 5 / 3 / 2 m, step size 0.5, 30 iterations) over VoxelGridCloud(r * 0.2) clouds,
// then pcl::GeneralizedIterativeClosestPoint (30 iterations, 2.0 m correspondence gate) over VoxelGridCloud(0.5 / 0.4)
// clouds, returning gicp.getFitnessScore().   SURVEY.md 8f rank 4.
//
// PARITY UNPINNED against PCL: PCL / FLANN / Eigen are third-party dependencies absent from /root/reference and from this
// container (catkin: pcl_ros, version unpinned; ROS Noetic => PCL 1.10.0, FLANN 1.9.1, Eigen 3.3.7 inferred).  What follows
// restates the PUBLISHED algorithms as PCL 1.10 implements them, from this author's reading of
//   pcl/registration/impl/ndt.hpp               P2D-NDT, Magnusson 2009 eq. 6.9-6.21, More-Thuente 1994 line search
//   pcl/filters/impl/voxel_grid_covariance.hpp  leaf Gaussians: >= 6 points, single-pass covariance, eigenvalue floor 0.01
//   pcl/registration/impl/gicp.hpp              Segal et al. 2009: 20-NN covariances (1, 1, 0.001), Mahalanobis metric
//   pcl/registration/bfgs.h                     GSL vector_bfgs2 (Fletcher's line search, rho 0.01 sigma 0.01 tau 9 / 0.05 / 0.5)
//   pcl/registration/impl/registration.hpp      align(), getFitnessScore()
// with the parameters the reference sets and PCL's defaults for the rest (outlier ratio 0.55, transformation epsilon 0.1 (NDT) /
// 5e-4 (GICP), rotation epsilon 2e-3, 20 inner BFGS iterations, gradient tolerance 1e-2).  The rotation derivatives are DERIVED
// (R = Rx Ry Rz for NDT, Rz Ry Rx for GICP) and checked against finite differences (tests/test_oracle_loop.py) rather than
// recalled sign by sign.  Choices where the library's arithmetic cannot be known here (documented, same in the product):
//   * SelfAdjointEigenSolver<3x3> -> cyclic Jacobi (flo::jacobi_svd3 on the symmetric matrix); JacobiSVD<6x6>::solve restated;
//   * FLANN radius / k-NN results in (distance, index) order; float point transforms as c0*x + (c1*y + (c2*z + c3)) (PCL's SSE form).
// ============================================================================
#pragma once
#include "flo_common.h"
#include "flo_kdtree.h"
#include "flo_linalg.h"
#include <map>
#include <cstdio>
#include <cstdlib>

namespace flo {
namespace loop {

typedef double Vec6[6];

// ---- small float / double helpers --------------------------------------------------------------------------------------------
struct M4f { float m[16]; };  // column-major
static inline M4f m4f_identity() { M4f r{}; for (int i = 0; i < 4; ++i) r.m[i * 5] = 1.f; return r; }
static inline M4f m4f_mul(const M4f& a, const M4f& b) {
    M4f r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.m[i + 4 * j] = ((a.m[i] * b.m[4 * j] + a.m[i + 4] * b.m[1 + 4 * j]) + a.m[i + 8] * b.m[2 + 4 * j]) + a.m[i + 12] * b.m[3 + 4 * j];
    return r;
}
static inline M4f m4f_rot_axis(int axis, float angle) {  // Eigen::AngleAxisf(angle, Unit{X,Y,Z}) as a 4x4
    M4f r = m4f_identity();
    const float c = std::cos(angle), s = std::sin(angle);
    const int a = (axis + 1) % 3, b = (axis + 2) % 3;
    r.m[a + 4 * a] = c; r.m[b + 4 * b] = c; r.m[b + 4 * a] = s; r.m[a + 4 * b] = -s;
    return r;
}
static inline M4f m4f_translation(float x, float y, float z) { M4f r = m4f_identity(); r.m[12] = x; r.m[13] = y; r.m[14] = z; return r; }
static inline M4f m4f_from_d(const double* T) { M4f r; for (int i = 0; i < 16; ++i) r.m[i] = float(T[i]); return r; }
// pcl::transformPointCloud (float, SSE form): c0 * x + (c1 * y + (c2 * z + c3))
static inline void xform_pt(const M4f& t, const float x, const float y, const float z, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = t.m[i] * x + (t.m[i + 4] * y + (t.m[i + 8] * z + t.m[i + 12]));
}
// Eigen 3.3 Matrix3f::eulerAngles(0, 1, 2)
static inline void euler_xyz(const M4f& t, float* e) {
    auto c = [&](int i, int j) { return t.m[i + 4 * j]; };
    const float pi = 3.14159265358979323846f;
    e[0] = std::atan2(c(1, 2), c(2, 2));
    const float c2 = std::sqrt(c(0, 0) * c(0, 0) + c(0, 1) * c(0, 1));
    if (e[0] > 0.f) {
        e[0] -= pi;  // (res[0] > 0 here)
        e[1] = std::atan2(-c(0, 2), -c2);
    } else {
        e[1] = std::atan2(-c(0, 2), c2);
    }
    const float s1 = std::sin(e[0]), c1 = std::cos(e[0]);
    e[2] = std::atan2(s1 * c(2, 0) - c1 * c(1, 0), c1 * c(1, 1) - s1 * c(2, 1));
    e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2];
}
// (Translation * AngleAxis(X) * AngleAxis(Y) * AngleAxis(Z)).matrix(), all float
static inline M4f pose_from_p(const double* p) {
    return m4f_mul(m4f_mul(m4f_mul(m4f_translation(float(p[0]), float(p[1]), float(p[2])), m4f_rot_axis(0, float(p[3]))), m4f_rot_axis(1, float(p[4]))),
                   m4f_rot_axis(2, float(p[5])));
}

// symmetric 3x3 eigen-decomposition, eigenvalues ascending (stand-in for SelfAdjointEigenSolver<Matrix3d>)
static inline void sym_eig3(const double* A, double* evals, double* evecs) {
    double U[9], S[3], V[9];
    jacobi_svd3(A, U, S, V);  // symmetric PSD-ish input: singular vectors = eigenvectors, sign from U^T V
    for (int k = 0; k < 3; ++k) {
        const double sgn = (U[3 * k] * V[3 * k] + U[3 * k + 1] * V[3 * k + 1]) + U[3 * k + 2] * V[3 * k + 2];
        const int dst = 2 - k;  // descending singular values -> ascending eigenvalues
        evals[dst] = sgn < 0.0 ? -S[k] : S[k];
        for (int i = 0; i < 3; ++i) evecs[i + 3 * dst] = V[i + 3 * k];
    }
    // re-sort ascending (a negative eigenvalue of large magnitude would be out of place)
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (evals[b] < evals[a]) { std::swap(evals[a], evals[b]); for (int i = 0; i < 3; ++i) std::swap(evecs[i + 3 * a], evecs[i + 3 * b]); }
}

// JacobiSVD<Matrix<double, N, N>>(A, ComputeFullU | ComputeFullV).solve(b): two-sided Jacobi, rank threshold eps * N * max sv
template <int N>
static inline void jacobi_svd_solve(const double* A, const double* b, double* x) {
    const double precision = 2.0 * std::numeric_limits<double>::epsilon(), consider_zero = std::numeric_limits<double>::min();
    double scale = 0.0;
    for (int i = 0; i < N * N; ++i) scale = std::max(scale, std::fabs(A[i]));
    if (scale == 0.0) scale = 1.0;
    double W[N * N], U[N * N], V[N * N];
    for (int i = 0; i < N * N; ++i) { W[i] = A[i] / scale; U[i] = V[i] = (i % (N + 1) == 0) ? 1.0 : 0.0; }
    double max_diag = 0.0;
    for (int i = 0; i < N; ++i) max_diag = std::max(max_diag, std::fabs(W[i + N * i]));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 1000) {
        finished = true;
        for (int p = 1; p < N; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = std::max(consider_zero, precision * max_diag);
                if (std::fabs(W[p + q * N]) > threshold || std::fabs(W[q + p * N]) > threshold) {
                    finished = false;
                    const double m00 = W[p + p * N], m01 = W[p + q * N], m10 = W[q + p * N], m11 = W[q + q * N];
                    JRot rot1;
                    const double t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < std::numeric_limits<double>::min()) { rot1.s = 0.0; rot1.c = 1.0; }
                    else { const double u = t / d, tmp = std::sqrt(1.0 + u * u); rot1.s = 1.0 / tmp; rot1.c = u / tmp; }
                    const double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11, n11 = -rot1.s * m01 + rot1.c * m11;
                    JRot jr;
                    make_jacobi(n00, n01, n11, jr);
                    const JRot jrt{jr.c, -jr.s};
                    const JRot jl{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
                    for (int k = 0; k < N; ++k) {  // W.applyOnTheLeft(p, q, j_left)
                        const double xi = W[p + k * N], yi = W[q + k * N];
                        W[p + k * N] = jl.c * xi + jl.s * yi;
                        W[q + k * N] = -jl.s * xi + jl.c * yi;
                    }
                    for (int k = 0; k < N; ++k) {  // U.applyOnTheRight(p, q, j_left.transpose())
                        const double xi = U[k + p * N], yi = U[k + q * N];
                        U[k + p * N] = jl.c * xi + jl.s * yi;
                        U[k + q * N] = -jl.s * xi + jl.c * yi;
                    }
                    for (int k = 0; k < N; ++k) {  // W.applyOnTheRight(p, q, j_right); V likewise
                        double xi = W[k + p * N], yi = W[k + q * N];
                        W[k + p * N] = jr.c * xi - jr.s * yi;
                        W[k + q * N] = jr.s * xi + jr.c * yi;
                        xi = V[k + p * N]; yi = V[k + q * N];
                        V[k + p * N] = jr.c * xi - jr.s * yi;
                        V[k + q * N] = jr.s * xi + jr.c * yi;
                    }
                    max_diag = std::max(max_diag, std::max(std::fabs(W[p + p * N]), std::fabs(W[q + q * N])));
                }
            }
    }
    double S[N];
    for (int i = 0; i < N; ++i) {
        const double a = W[i + i * N];
        S[i] = std::fabs(a) * scale;
        if (a < 0.0) for (int k = 0; k < N; ++k) U[k + i * N] = -U[k + i * N];
    }
    double smax = 0.0;
    for (int i = 0; i < N; ++i) smax = std::max(smax, S[i]);
    const double thr = std::numeric_limits<double>::epsilon() * N * smax;  // JacobiSVD::rank(): default threshold
    for (int i = 0; i < N; ++i) x[i] = 0.0;
    for (int k = 0; k < N; ++k) {
        if (!(S[k] > thr)) continue;
        double ub = 0.0;
        for (int i = 0; i < N; ++i) ub += U[i + k * N] * b[i];
        const double c = ub / S[k];
        for (int i = 0; i < N; ++i) x[i] += V[i + k * N] * c;
    }
}

// ---- pcl::VoxelGridCovariance (leaf = resolution, min_points_per_voxel 6, min_covar_eigvalue_mult 0.01), searchable -----------
struct Leaf {
    int nr = 0;
    double mean[3] = {0, 0, 0}, cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, icov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float centroid[3] = {0, 0, 0};
    double sum[3] = {0, 0, 0}, xx[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};
struct TargetCells {
    float leaf_size = 1.f, inv = 1.f;
    int min_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0};
    std::map<int, Leaf> leaves;
    std::vector<int> searchable;       // leaf index of every voxel centroid, in map (= ascending index) order
    std::vector<float> centroids;      // xyz of those centroids (the cloud FLANN indexes)
    std::vector<const Leaf*> leaf_of;  // same order
    bool ok = false;
    void build(const Cloud& in, float resolution) {
        leaves.clear(); searchable.clear(); centroids.clear(); leaf_of.clear();
        ok = false;
        leaf_size = resolution;
        inv = 1.0f / resolution;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool any = false;
        for (const P4& p : in) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            any = true;
            mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
            mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
            mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
        }
        if (!any) return;
        const int64_t dx = int64_t((mx[0] - mn[0]) * inv) + 1, dy = int64_t((mx[1] - mn[1]) * inv) + 1, dz = int64_t((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > int64_t(std::numeric_limits<int32_t>::max())) return;
        for (int a = 0; a < 3; ++a) { min_b[a] = int(std::floor(mn[a] * inv)); div_b[a] = int(std::floor(mx[a] * inv)) - min_b[a] + 1; }
        const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
        for (const P4& p : in) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            const int i0 = int(std::floor(p.x * inv) - float(min_b[0])), i1 = int(std::floor(p.y * inv) - float(min_b[1])), i2 = int(std::floor(p.z * inv) - float(min_b[2]));
            Leaf& l = leaves[i0 * mul[0] + i1 * mul[1] + i2 * mul[2]];
            const double q[3] = {double(p.x), double(p.y), double(p.z)};
            for (int a = 0; a < 3; ++a) l.sum[a] += q[a];
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) l.xx[r + 3 * c] += q[r] * q[c];
            l.centroid[0] += p.x; l.centroid[1] += p.y; l.centroid[2] += p.z;
            ++l.nr;
        }
        for (auto& kv : leaves) {
            Leaf& l = kv.second;
            const int n = l.nr;
            for (int a = 0; a < 3; ++a) { l.centroid[a] /= float(n); l.mean[a] = l.sum[a] / n; }
            if (n < 6) continue;
            searchable.push_back(kv.first);
            // single-pass covariance: (sum xx^T - 2 (sum x) mean^T) / n + mean mean^T, then * (n - 1) / n
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) l.cov[r + 3 * c] = (l.xx[r + 3 * c] - 2.0 * (l.sum[r] * l.mean[c])) / n + l.mean[r] * l.mean[c];
            for (int i = 0; i < 9; ++i) l.cov[i] *= (n - 1.0) / n;
            double ev[3], evec[9];
            sym_eig3(l.cov, ev, evec);
            if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) { l.nr = -1; continue; }
            const double floor_ev = 0.01 * ev[2];
            if (ev[0] < floor_ev) {
                ev[0] = floor_ev;
                if (ev[1] < floor_ev) ev[1] = floor_ev;
                double vinv[9], tmp[9];
                inverse3(evec, vinv);
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) tmp[r + 3 * c] = evec[r + 3 * c] * ev[c];  // evecs * diag
                mat3_mul(tmp, vinv, l.cov);
            }
            inverse3(l.cov, l.icov);
            double mxc = -INFINITY, mnc = INFINITY;
            for (int i = 0; i < 9; ++i) { mxc = std::max(mxc, l.icov[i]); mnc = std::min(mnc, l.icov[i]); }
            if (mxc == double(INFINITY) || mnc == -double(INFINITY)) l.nr = -1;
        }
        for (int idx : searchable) {
            const Leaf& l = leaves[idx];
            centroids.push_back(l.centroid[0]); centroids.push_back(l.centroid[1]); centroids.push_back(l.centroid[2]);
            leaf_of.push_back(&l);
        }
        ok = !searchable.empty();
    }
    // radiusSearch(point, radius): voxel centroids with squared float distance < radius^2, ascending (distance, index)
    void radius_search(const float* q, float radius, std::vector<int>& out) const {
        out.clear();
        const float r2 = radius * radius;
        struct H { float d2; int i; };
        H hits[64];
        int nh = 0;
        // every centroid lies inside its own leaf, so hits are confined to the 27 leaves around the query's leaf
        const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
        const int c0 = int(std::floor(q[0] * inv) - float(min_b[0])), c1 = int(std::floor(q[1] * inv) - float(min_b[1])), c2 = int(std::floor(q[2] * inv) - float(min_b[2]));
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = c0 + dx, y = c1 + dy, z = c2 + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= div_b[0] || y >= div_b[1] || z >= div_b[2]) continue;
                    const int idx = x * mul[0] + y * mul[1] + z * mul[2];
                    const auto it = std::lower_bound(searchable.begin(), searchable.end(), idx);
                    if (it == searchable.end() || *it != idx) continue;
                    const int k = int(it - searchable.begin());
                    const float d2 = l2_simple(q, &centroids[3 * size_t(k)]);
                    if (d2 < r2) hits[nh++] = H{d2, k};
                }
        std::sort(hits, hits + nh, [](const H& a, const H& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.i < b.i); });
        for (int k = 0; k < nh; ++k) out.push_back(hits[k].i);
    }
};

// ---- pcl::NormalDistributionsTransform ----------------------------------------------------------------------------------------
struct NdtStats { int iterations = 0, evaluations = 0; double score = 0.0; bool converged = false; };
struct Ndt {
    float resolution = 1.0f;
    double step_size = 0.1, outlier_ratio = 0.55, transformation_epsilon = 0.1;
    int max_iterations = 35;
    const Cloud* source = nullptr;
    TargetCells cells;
    double gauss_d1 = 0, gauss_d2 = 0;
    double j_ang[8][3], h_ang[15][3];
    M4f final_transformation = m4f_identity();
    NdtStats stats;

    void set_target(const Cloud& target) { cells.build(target, resolution); }

    // d(Rx Ry Rz x)/d(angles) and the second derivatives: the rows of Magnusson eq. 6.19 / 6.21, DERIVED from R = Rx(a) Ry(b) Rz(c)
    void angle_derivatives(const double* p, bool hessian = true) {
        double cx, cy, cz, sx, sy, sz;
        if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
        if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
        if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
        // R = [ cy cz, -cy sz, sy ; cx sz + sx sy cz, cx cz - sx sy sz, -sx cy ; sx sz - cx sy cz, sx cz + cx sy sz, cx cy ]
        auto set = [](double* d, double a, double b, double c) { d[0] = a; d[1] = b; d[2] = c; };
        set(j_ang[0], -sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy);  // a: d row1 / d rx
        set(j_ang[1], cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy);    // b: d row2 / d rx
        set(j_ang[2], -sy * cz, sy * sz, cy);                                        // c: d row0 / d ry
        set(j_ang[3], sx * cy * cz, -sx * cy * sz, sx * sy);                         // d: d row1 / d ry
        set(j_ang[4], -cx * cy * cz, cx * cy * sz, -cx * sy);                        // e: d row2 / d ry
        set(j_ang[5], -cy * sz, -cy * cz, 0.0);                                      // f: d row0 / d rz
        set(j_ang[6], cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0.0);         // g: d row1 / d rz
        set(j_ang[7], sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0.0);          // h: d row2 / d rz
        if (!hessian) return;
        set(h_ang[0], -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy);    // a2: d2 row1 / d rx2
        set(h_ang[1], -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy);   // a3: d2 row2 / d rx2
        set(h_ang[2], cx * cy * cz, -cx * cy * sz, cx * sy);                         // b2: d2 row1 / d rx d ry
        set(h_ang[3], sx * cy * cz, -sx * cy * sz, sx * sy);                         // b3: d2 row2 / d rx d ry
        set(h_ang[4], -sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0.0);         // c2: d2 row1 / d rx d rz
        set(h_ang[5], cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0.0);         // c3: d2 row2 / d rx d rz
        set(h_ang[6], -cy * cz, cy * sz, -sy);                                       // d1: d2 row0 / d ry2
        set(h_ang[7], -sx * sy * cz, sx * sy * sz, sx * cy);                         // d2: d2 row1 / d ry2
        set(h_ang[8], cx * sy * cz, -cx * sy * sz, -cx * cy);                        // d3: d2 row2 / d ry2
        set(h_ang[9], sy * sz, sy * cz, 0.0);                                        // e1: d2 row0 / d ry d rz
        set(h_ang[10], -sx * cy * sz, -sx * cy * cz, 0.0);                           // e2: d2 row1 / d ry d rz
        set(h_ang[11], cx * cy * sz, cx * cy * cz, 0.0);                             // e3: d2 row2 / d ry d rz
        set(h_ang[12], -cy * cz, cy * sz, 0.0);                                      // f1: d2 row0 / d rz2
        set(h_ang[13], -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0.0);       // f2: d2 row1 / d rz2
        set(h_ang[14], -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0.0);       // f3: d2 row2 / d rz2
    }

    // computeDerivatives (+ computeHessian when only_hessian): sums over every source point and every neighbouring leaf
    double derivatives(double* grad /*6*/, double* hess /*36 col-major*/, const M4f& T, const double* p, bool with_hessian, bool only_hessian = false) {
        for (int i = 0; i < 6; ++i) grad[i] = 0.0;
        if (with_hessian) for (int i = 0; i < 36; ++i) hess[i] = 0.0;
        double score = 0.0;
        angle_derivatives(p, with_hessian);
        std::vector<int> nb;
        ++stats.evaluations;
        for (const P4& sp : *source) {
            float xt[3];
            xform_pt(T, sp.x, sp.y, sp.z, xt);
            cells.radius_search(xt, resolution, nb);
            if (nb.empty()) continue;
            const double x[3] = {double(sp.x), double(sp.y), double(sp.z)};
            // computePointDerivatives: point_gradient_ (3 x 6), point_hessian_ (18 x 6)
            double pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
            auto dot = [&](const double* v) { return (x[0] * v[0] + x[1] * v[1]) + x[2] * v[2]; };
            pg[1][3] = dot(j_ang[0]); pg[2][3] = dot(j_ang[1]);
            pg[0][4] = dot(j_ang[2]); pg[1][4] = dot(j_ang[3]); pg[2][4] = dot(j_ang[4]);
            pg[0][5] = dot(j_ang[5]); pg[1][5] = dot(j_ang[6]); pg[2][5] = dot(j_ang[7]);
            double ph[6][6][3];
            if (with_hessian) {
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) ph[i][j][0] = ph[i][j][1] = ph[i][j][2] = 0.0;
                const double a[3] = {0.0, dot(h_ang[0]), dot(h_ang[1])}, b[3] = {0.0, dot(h_ang[2]), dot(h_ang[3])}, c[3] = {0.0, dot(h_ang[4]), dot(h_ang[5])};
                const double d[3] = {dot(h_ang[6]), dot(h_ang[7]), dot(h_ang[8])}, e[3] = {dot(h_ang[9]), dot(h_ang[10]), dot(h_ang[11])};
                const double f[3] = {dot(h_ang[12]), dot(h_ang[13]), dot(h_ang[14])};
                for (int k = 0; k < 3; ++k) {
                    ph[3][3][k] = a[k]; ph[3][4][k] = ph[4][3][k] = b[k]; ph[3][5][k] = ph[5][3][k] = c[k];
                    ph[4][4][k] = d[k]; ph[4][5][k] = ph[5][4][k] = e[k]; ph[5][5][k] = f[k];
                }
            }
            for (int k : nb) {
                const Leaf& l = *cells.leaf_of[size_t(k)];
                const double xq[3] = {double(xt[0]) - l.mean[0], double(xt[1]) - l.mean[1], double(xt[2]) - l.mean[2]};
                const double* ci = l.icov;
                auto cmul = [&](const double* v, double* o) {  // c_inv * v
                    for (int r = 0; r < 3; ++r) o[r] = (ci[r] * v[0] + ci[r + 3] * v[1]) + ci[r + 6] * v[2];
                };
                double cx[3];
                cmul(xq, cx);
                double e_x = std::exp(-gauss_d2 * ((xq[0] * cx[0] + xq[1] * cx[1]) + xq[2] * cx[2]) / 2.0);
                const double score_inc = -gauss_d1 * e_x;
                e_x = gauss_d2 * e_x;
                if (e_x > 1.0 || e_x < 0.0 || e_x != e_x) continue;  // updateDerivatives returns 0
                e_x *= gauss_d1;
                double cg[6][3], xcg[6];
                for (int i = 0; i < 6; ++i) {
                    const double col[3] = {pg[0][i], pg[1][i], pg[2][i]};
                    cmul(col, cg[i]);
                    xcg[i] = (xq[0] * cg[i][0] + xq[1] * cg[i][1]) + xq[2] * cg[i][2];
                    if (!only_hessian) grad[i] += xcg[i] * e_x;
                }
                if (with_hessian)
                    for (int i = 0; i < 6; ++i)
                        for (int j = 0; j < 6; ++j) {
                            double chv[3];
                            cmul(ph[i][j], chv);
                            const double t2 = (xq[0] * chv[0] + xq[1] * chv[1]) + xq[2] * chv[2];
                            const double t3 = (pg[0][j] * cg[i][0] + pg[1][j] * cg[i][1]) + pg[2][j] * cg[i][2];
                            hess[i + 6 * j] += e_x * ((-gauss_d2 * xcg[i] * xcg[j] + t2) + t3);
                        }
                score += score_inc;
            }
        }
        return score;
    }

    // More-Thuente helpers (ndt.hpp auxilaryFunction_PsiMT / dPsiMT, updateIntervalMT, trialValueSelectionMT)
    static double psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
    static double dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
    static bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
        if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
        if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
        if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
        return true;
    }
    static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
        if (f_t > f_l) {
            const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
            const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
            const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
            return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
        }
        if (g_t * g_l < 0) {
            const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
            const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
            const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
            return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
        }
        if (std::fabs(g_t) <= std::fabs(g_l)) {
            const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
            const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
            const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
            const double a_next = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
            return a_t > a_l ? std::min(a_t + 0.66 * (a_u - a_t), a_next) : std::max(a_t + 0.66 * (a_u - a_t), a_next);
        }
        const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = std::sqrt(z * z - g_t * g_u);
        return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
    }
    double step_length_mt(const double* x, double* step_dir, double step_init, double step_max, double step_min, double& score, double* grad, double* hess) {
        const double phi_0 = -score;
        double d_phi_0 = 0.0;
        for (int i = 0; i < 6; ++i) d_phi_0 += grad[i] * step_dir[i];
        d_phi_0 = -d_phi_0;
        if (d_phi_0 >= 0) {
            if (d_phi_0 == 0) return 0;
            d_phi_0 *= -1;
            for (int i = 0; i < 6; ++i) step_dir[i] *= -1;
        }
        const int max_step_iterations = 10;
        int step_iterations = 0;
        const double mu = 1.e-4, nu = 0.9;
        double a_l = 0, a_u = 0;
        double f_l = psi(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi(d_phi_0, d_phi_0, mu);
        double f_u = psi(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi(d_phi_0, d_phi_0, mu);
        bool interval_converged = (step_max - step_min) < 0, open_interval = true;
        double a_t = std::max(std::min(step_init, step_max), step_min);
        double x_t[6];
        for (int i = 0; i < 6; ++i) x_t[i] = x[i] + step_dir[i] * a_t;
        final_transformation = pose_from_p(x_t);
        score = derivatives(grad, hess, final_transformation, x_t, true);
        double phi_t = -score, d_phi_t = 0.0;
        for (int i = 0; i < 6; ++i) d_phi_t += grad[i] * step_dir[i];
        d_phi_t = -d_phi_t;
        double psi_t = psi(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi(d_phi_t, d_phi_0, mu);
        while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
            a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
            a_t = std::max(std::min(a_t, step_max), step_min);
            for (int i = 0; i < 6; ++i) x_t[i] = x[i] + step_dir[i] * a_t;
            final_transformation = pose_from_p(x_t);
            score = derivatives(grad, hess, final_transformation, x_t, false);
            phi_t = -score;
            d_phi_t = 0.0;
            for (int i = 0; i < 6; ++i) d_phi_t += grad[i] * step_dir[i];
            d_phi_t = -d_phi_t;
            psi_t = psi(a_t, phi_t, phi_0, d_phi_0, mu);
            d_psi_t = dpsi(d_phi_t, d_phi_0, mu);
            if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
                open_interval = false;
                f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
                f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
            }
            interval_converged = open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                               : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
            ++step_iterations;
        }
        if (step_iterations) {  // computeHessian at x_t (the gradient is current already)
            double gdummy[6];
            derivatives(gdummy, hess, final_transformation, x_t, true, /*only_hessian=*/true);
        }
        return a_t;
    }
    // align(output, guess): computeTransformation
    M4f align(const Cloud& src, const M4f& guess) {
        source = &src;
        stats = NdtStats();
        const double c1 = 10.0 * (1.0 - outlier_ratio), c2 = outlier_ratio / std::pow(double(resolution), 3), d3 = -std::log(c2);
        gauss_d1 = -std::log(c1 + c2) - d3;
        gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / gauss_d1);
        final_transformation = guess;
        if (!cells.ok || src.empty()) return final_transformation;
        float e[3];
        euler_xyz(final_transformation, e);
        double p[6] = {double(final_transformation.m[12]), double(final_transformation.m[13]), double(final_transformation.m[14]), double(e[0]), double(e[1]), double(e[2])};
        double grad[6], hess[36], delta_p[6];
        double score = derivatives(grad, hess, final_transformation, p, true);
        bool converged = false;
        int nr = 0;
        while (!converged) {
            double neg[6];
            for (int i = 0; i < 6; ++i) neg[i] = -grad[i];
            jacobi_svd_solve<6>(hess, neg, delta_p);
            double nrm = 0.0;
            for (int i = 0; i < 6; ++i) nrm += delta_p[i] * delta_p[i];
            nrm = std::sqrt(nrm);
            if (nrm == 0 || nrm != nrm) { stats.converged = nrm == nrm; break; }
            for (int i = 0; i < 6; ++i) delta_p[i] /= nrm;
            nrm = step_length_mt(p, delta_p, nrm, step_size, transformation_epsilon / 2, score, grad, hess);
            for (int i = 0; i < 6; ++i) { delta_p[i] *= nrm; p[i] += delta_p[i]; }
            if (nr > max_iterations || (nr && std::fabs(nrm) < transformation_epsilon)) { converged = true; stats.converged = true; }
            ++nr;
        }
        stats.iterations = nr;
        stats.score = src.empty() ? 0.0 : score / double(src.size());
        return final_transformation;
    }
};

// ---- pcl::GeneralizedIterativeClosestPoint ----------------------------------------------------------------------------------
struct GicpStats { int iterations = 0, inner_total = 0, evaluations = 0, correspondences = 0; bool failed = false; };
struct Gicp {
    int k_correspondences = 20, max_iterations = 200, max_inner_iterations = 20;
    double gicp_epsilon = 0.001, rotation_epsilon = 2e-3, transformation_epsilon = 5e-4, corr_dist_threshold = 5.0;
    const Cloud* target = nullptr;
    Cloud moved;  // `output`: the source transformed by the guess
    KdTree tgt_tree;
    std::vector<double> cov_src, cov_tgt, mahal;  // 9 doubles per point
    std::vector<int> idx_src, idx_tgt;
    GicpStats stats;

    static void covariances(const Cloud& c, int k, double eps, std::vector<double>& out) {
        out.assign(c.size() * 9, 0.0);
        if (k > int(c.size())) return;
        KdTree tree;
        tree.Build(&c[0].x, c.size(), 4);
        std::vector<KdTree::Hit> hits; hits.resize(size_t(k));
#pragma omp parallel for firstprivate(hits) schedule(dynamic, 256)
        for (long long i = 0; i < (long long)c.size(); ++i) {
            const float q[3] = {c[size_t(i)].x, c[size_t(i)].y, c[size_t(i)].z};
            const int n = tree.Knn(q, k, hits.data());
            double mean[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < n; ++j) {
                const P4& pt = c[size_t(hits[size_t(j)].idx)];
                mean[0] += pt.x; mean[1] += pt.y; mean[2] += pt.z;
                cov[0] += pt.x * pt.x;  // float products, double sums
                cov[1] += pt.y * pt.x; cov[4] += pt.y * pt.y;
                cov[2] += pt.z * pt.x; cov[5] += pt.z * pt.y; cov[8] += pt.z * pt.z;
            }
            for (int a = 0; a < 3; ++a) mean[a] /= double(k);
            for (int r = 0; r < 3; ++r)
                for (int l = 0; l <= r; ++l) {
                    double v = cov[r + 3 * l] / double(k);
                    v -= mean[r] * mean[l];
                    cov[r + 3 * l] = cov[l + 3 * r] = v;
                }
            double U[9], S[3], V[9];
            jacobi_svd3(cov, U, S, V);
            double* o = &out[size_t(i) * 9];
            for (int q9 = 0; q9 < 9; ++q9) o[q9] = 0.0;
            for (int kk = 0; kk < 3; ++kk) {
                const double v = kk == 2 ? eps : 1.0;
                for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) o[r + 3 * cc] += v * U[r + 3 * kk] * U[cc + 3 * kk];
            }
        }
    }
    // R = Rz(psi) Ry(theta) Rx(phi) applied on the left of t's rotation, translation added (applyState), float
    static void apply_state(M4f& t, const double* x) {
        const M4f R = m4f_mul(m4f_mul(m4f_rot_axis(2, float(x[5])), m4f_rot_axis(1, float(x[4]))), m4f_rot_axis(0, float(x[3])));
        M4f o = t;
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) o.m[i + 4 * j] = (R.m[i] * t.m[4 * j] + R.m[i + 4] * t.m[1 + 4 * j]) + R.m[i + 8] * t.m[2 + 4 * j];
        o.m[12] = t.m[12] + float(x[0]); o.m[13] = t.m[13] + float(x[1]); o.m[14] = t.m[14] + float(x[2]);
        t = o;
    }
    // f, df (OptimizationFunctorWithIndices): base_transformation_ = identity
    void fdf(const double* x, double* f, double* g) {
        ++stats.evaluations;
        M4f T = m4f_identity();
        apply_state(T, x);
        const int m = int(idx_src.size());
        double fs = 0.0, gt[3] = {0, 0, 0}, Racc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < m; ++i) {
            const P4& ps = moved[size_t(idx_src[size_t(i)])];
            const P4& pt = (*target)[size_t(idx_tgt[size_t(i)])];
            float pp[3];
            xform_pt(T, ps.x, ps.y, ps.z, pp);
            const double res[3] = {double(pp[0] - pt.x), double(pp[1] - pt.y), double(pp[2] - pt.z)};
            const double* M = &mahal[size_t(idx_src[size_t(i)]) * 9];
            double tmp[3];
            for (int r = 0; r < 3; ++r) tmp[r] = (M[r] * res[0] + M[r + 3] * res[1]) + M[r + 6] * res[2];
            fs += (res[0] * tmp[0] + res[1] * tmp[1]) + res[2] * tmp[2];
            if (g) {
                for (int a = 0; a < 3; ++a) gt[a] += tmp[a];
                const double p3[3] = {double(ps.x), double(ps.y), double(ps.z)};  // base_transformation_ * p_src
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Racc[r + 3 * c] += p3[r] * tmp[c];
            }
        }
        if (f) *f = fs / m;
        if (!g) return;
        for (int a = 0; a < 3; ++a) g[a] = gt[a] * (2.0 / m);
        for (int q = 0; q < 9; ++q) Racc[q] *= 2.0 / m;
        // computeRDerivative: d(Rz Ry Rx)/d(phi, theta, psi), g[3 + k] = sum_ij dR_k(i, j) * Racc(j, i)
        const double phi = x[3], theta = x[4], psi = x[5];
        const double cphi = std::cos(phi), sphi = std::sin(phi), cth = std::cos(theta), sth = std::sin(theta), cpsi = std::cos(psi), spsi = std::sin(psi);
        double dphi[9], dth[9], dpsi_[9];  // column-major
        auto S = [](double* d, int i, int j, double v) { d[i + 3 * j] = v; };
        S(dphi, 0, 0, 0.); S(dphi, 1, 0, 0.); S(dphi, 2, 0, 0.);
        S(dphi, 0, 1, sphi * spsi + cphi * cpsi * sth); S(dphi, 1, 1, -cpsi * sphi + cphi * spsi * sth); S(dphi, 2, 1, cphi * cth);
        S(dphi, 0, 2, cphi * spsi - cpsi * sphi * sth); S(dphi, 1, 2, -cphi * cpsi - sphi * spsi * sth); S(dphi, 2, 2, -cth * sphi);
        S(dth, 0, 0, -cpsi * sth); S(dth, 1, 0, -spsi * sth); S(dth, 2, 0, -cth);
        S(dth, 0, 1, cpsi * cth * sphi); S(dth, 1, 1, cth * sphi * spsi); S(dth, 2, 1, -sphi * sth);
        S(dth, 0, 2, cphi * cpsi * cth); S(dth, 1, 2, cphi * cth * spsi); S(dth, 2, 2, -cphi * sth);
        S(dpsi_, 0, 0, -cth * spsi); S(dpsi_, 1, 0, cpsi * cth); S(dpsi_, 2, 0, 0.);
        S(dpsi_, 0, 1, -cphi * cpsi - sphi * spsi * sth); S(dpsi_, 1, 1, -cphi * spsi + cpsi * sphi * sth); S(dpsi_, 2, 1, 0.);
        S(dpsi_, 0, 2, cpsi * sphi - cphi * spsi * sth); S(dpsi_, 1, 2, sphi * spsi + cphi * cpsi * sth); S(dpsi_, 2, 2, 0.);
        auto inner = [&](const double* d) { double r = 0.0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r += d[j + 3 * i] * Racc[i + 3 * j]; return r; };
        g[3] = inner(dphi); g[4] = inner(dth); g[5] = inner(dpsi_);
    }

    // pcl/registration/bfgs.h (GSL vector_bfgs2 + Fletcher's line search)
    struct Bfgs {
        Gicp* host;
        double x0[6], g0[6], p[6], x_alpha[6], g_alpha[6], dx[6], gradient[6];
        double f = 0, f_alpha = 0, df_alpha = 0, g0norm = 0, pnorm = 0, fp0 = 0, delta_f = 0;
        double x_key = 0, f_key = 0, g_key = 0, df_key = 0;
        static double nrm(const double* v) { double s = 0; for (int i = 0; i < 6; ++i) s += v[i] * v[i]; return std::sqrt(s); }
        static double dot6(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }
        void move_to(double a) { if (a == x_key) return; for (int i = 0; i < 6; ++i) x_alpha[i] = x0[i] + a * p[i]; x_key = a; }
        double slope() const { return dot6(g_alpha, p); }
        double apply_f(double a) { if (a == f_key) return f_alpha; move_to(a); host->fdf(x_alpha, &f_alpha, nullptr); f_key = a; return f_alpha; }
        double apply_df(double a) {
            if (a == df_key) return df_alpha;
            move_to(a);
            if (a != g_key) { double ft; host->fdf(x_alpha, &ft, g_alpha); g_key = a; }
            df_alpha = slope(); df_key = a;
            return df_alpha;
        }
        void apply_fdf(double a, double& fv, double& dfv) {
            if (a == f_key && a == df_key) { fv = f_alpha; dfv = df_alpha; return; }
            if (a == f_key || a == df_key) { fv = apply_f(a); dfv = apply_df(a); return; }
            move_to(a);
            host->fdf(x_alpha, &f_alpha, g_alpha);
            f_key = a; g_key = a;
            df_alpha = slope(); df_key = a;
            fv = f_alpha; dfv = df_alpha;
        }
        void change_direction() {
            for (int i = 0; i < 6; ++i) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }
            x_key = 0; f_key = 0; g_key = 0;
            df_alpha = slope(); df_key = 0;
        }
        void init(const double* x) {
            delta_f = 0;
            for (int i = 0; i < 6; ++i) dx[i] = 0;
            host->fdf(x, &f, gradient);
            for (int i = 0; i < 6; ++i) { x0[i] = x[i]; g0[i] = gradient[i]; }
            g0norm = nrm(g0);
            for (int i = 0; i < 6; ++i) p[i] = gradient[i] * (-1.0 / g0norm);
            pnorm = nrm(p);
            fp0 = -g0norm;
            for (int i = 0; i < 6; ++i) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }
            x_key = 0; f_alpha = f; f_key = 0; g_key = 0;
            df_alpha = slope(); df_key = 0;
        }
        static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
        static void check_extremum(double c0, double c1, double c2, double c3, double z, double& zmin, double& fmin) {
            const double y = cubic(c0, c1, c2, c3, z);
            if (y < fmin) { zmin = z; fmin = y; }
        }
        static int solve_quadratic(double a, double b, double c, double& r0, double& r1) {
            const double disc = b * b - 4 * a * c;
            if (a == 0) { if (b == 0) return 0; r0 = -c / b; return 1; }
            if (disc > 0) {
                if (b == 0) { const double r = std::fabs(0.5 * std::sqrt(disc) / a); r0 = -r; r1 = r; }
                else {
                    const double sgnb = b > 0 ? 1 : -1, temp = -0.5 * (b + sgnb * std::sqrt(disc)), q1 = temp / a, q2 = c / temp;
                    if (q1 < q2) { r0 = q1; r1 = q2; } else { r0 = q2; r1 = q1; }
                }
                return 2;
            }
            if (disc == 0) { r0 = r1 = -0.5 * b / a; return 2; }
            return 0;
        }
        static double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
            const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0)), fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0)), c = 2 * (f1 - f0 - fp0);
            double zmin = zl, fmin = fl;
            if (fh < fmin) { zmin = zh; fmin = fh; }
            if (c > 0) {
                const double z = -fp0 / c;
                if (z > zl && z < zh) { const double fz = f0 + z * (fp0 + z * (f1 - f0 - fp0)); if (fz < fmin) { zmin = z; fmin = fz; } }
            }
            return zmin;
        }
        static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
            const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1, xi = fp0 + fp1 - 2 * (f1 - f0);
            const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
            double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0 = 0, z1 = 0;
            check_extremum(c0, c1, c2, c3, zh, zmin, fmin);
            const int n = solve_quadratic(3 * c3, 2 * c2, c1, z0, z1);
            if (n == 2) {
                if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, zmin, fmin);
                if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, zmin, fmin);
            } else if (n == 1) {
                if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, zmin, fmin);
            }
            return zmin;
        }
        static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
            double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
            if (ymin > ymax) std::swap(ymin, ymax);
            double y;
            if (order > 2 && !(fpb != fpb) && fpb != std::numeric_limits<double>::infinity()) y = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);
            else y = interp_quad(fa, fpa * (b - a), fb, ymin, ymax);
            return a + y * (b - a);
        }
        // 0 = Success, 1 = NoProgress
        int line_search(double rho, double sigma, double tau1, double tau2, double tau3, int order, double alpha1, double& alpha_new) {
            double f0v, fp0v, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
            double alpha = alpha1, alpha_prev = 0.0, a, b, fa, fb, fpa, fpb;
            int i = 0;
            apply_fdf(0.0, f0v, fp0v);
            falpha_prev = f0v; fpalpha_prev = fp0v;
            a = 0.0; b = alpha; fa = f0v; fb = 0.0; fpa = fp0v; fpb = 0.0;
            const double nan = std::numeric_limits<double>::quiet_NaN();
            while (i++ < 100) {
                falpha = apply_f(alpha);
                if (falpha > f0v + alpha * rho * fp0v || falpha >= falpha_prev) { a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev; b = alpha; fb = falpha; fpb = nan; break; }
                fpalpha = apply_df(alpha);
                if (std::fabs(fpalpha) <= -sigma * fp0v) { alpha_new = alpha; return 0; }
                if (fpalpha >= 0) { a = alpha; fa = falpha; fpa = fpalpha; b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev; break; }
                delta = alpha - alpha_prev;
                alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order);
                alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
            }
            while (i++ < 100) {
                delta = b - a;
                alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
                falpha = apply_f(alpha);
                if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return 1;  // roundoff prevents progress
                if (falpha > f0v + rho * alpha * fp0v || falpha >= fa) { b = alpha; fb = falpha; fpb = nan; }
                else {
                    fpalpha = apply_df(alpha);
                    if (std::fabs(fpalpha) <= -sigma * fp0v) { alpha_new = alpha; return 0; }
                    if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; fb = fa; fpb = fpa; a = alpha; fa = falpha; fpa = fpalpha; }
                    else { a = alpha; fa = falpha; fpa = fpalpha; }
                }
            }
            return 0;
        }
        int one_step(double* x) {
            double alpha = 0.0, alpha1;
            const double f0v = f;
            if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) { for (int i = 0; i < 6; ++i) dx[i] = 0; return 1; }
            if (delta_f < 0) {
                const double del = std::max(-delta_f, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0v));
                alpha1 = std::min(1.0, 2.0 * del / (-fp0));
            } else alpha1 = 1.0;  // |parameters.step_size|
            const int status = line_search(0.01, 0.01, 9, 0.05, 0.5, 3, alpha1, alpha);
            if (status != 0) return status;
            apply_fdf(alpha, f_alpha, df_alpha);  // updatePosition
            for (int i = 0; i < 6; ++i) { x[i] = x_alpha[i]; gradient[i] = g_alpha[i]; }
            f = f_alpha;
            delta_f = f - f0v;
            double dx0[6], dg0[6];
            for (int i = 0; i < 6; ++i) { dx0[i] = x[i] - x0[i]; dx[i] = dx0[i]; dg0[i] = gradient[i] - g0[i]; }
            const double dxg = dot6(dx0, gradient), dgg = dot6(dg0, gradient), dxdg = dot6(dx0, dg0), dgnorm = nrm(dg0);
            double A = 0, B = 0;
            if (dxdg != 0) { B = dxg / dxdg; A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg; }
            for (int i = 0; i < 6; ++i) p[i] = (-A * dx0[i] + gradient[i]) + -B * dg0[i];
            for (int i = 0; i < 6; ++i) { g0[i] = gradient[i]; x0[i] = x[i]; }
            g0norm = nrm(g0);
            pnorm = nrm(p);
            const double dir = dot6(p, gradient) > 0 ? -1.0 : 1.0;
            for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
            pnorm = nrm(p);
            fp0 = dot6(p, g0);
            change_direction();
            return 0;
        }
    };

    // estimateRigidTransformationBFGS; false = SolverDidntConvergeException / NotEnoughPointsException
    bool estimate(M4f& transformation) {
        if (idx_src.size() < 4) return false;
        double x[6];
        x[0] = transformation.m[12]; x[1] = transformation.m[13]; x[2] = transformation.m[14];
        x[3] = std::atan2(double(transformation.m[2 + 4 * 1]), double(transformation.m[2 + 4 * 2]));
        x[4] = std::asin(-double(transformation.m[2 + 4 * 0]));
        x[5] = std::atan2(double(transformation.m[1 + 4 * 0]), double(transformation.m[0]));
        Bfgs b;
        b.host = this;
        b.init(x);
        int inner = 0, result;
        do {
            ++inner;
            result = b.one_step(x);
            if (result) break;
            result = Bfgs::nrm(b.gradient) < 1e-2 ? 0 : -1;  // testGradient: Success / Running
        } while (result == -1 && inner < max_inner_iterations);
        stats.inner_total += inner;
        if (result == 1 || result == 0 || inner == max_inner_iterations) {
            transformation = m4f_identity();
            apply_state(transformation, x);
            return true;
        }
        return false;
    }

    M4f align(const Cloud& src, const Cloud& tgt, const M4f& guess) {
        stats = GicpStats();
        target = &tgt;
        tgt_tree.Build(&tgt[0].x, tgt.size(), 4);
        covariances(tgt, k_correspondences, gicp_epsilon, cov_tgt);
        covariances(src, k_correspondences, gicp_epsilon, cov_src);
        const size_t N = src.size();
        mahal.assign(N * 9, 0.0);
        for (size_t i = 0; i < N; ++i) mahal[9 * i] = mahal[9 * i + 4] = mahal[9 * i + 8] = 1.0;
        moved.resize(N);
        for (size_t i = 0; i < N; ++i) { float o[3]; xform_pt(guess, src[i].x, src[i].y, src[i].z, o); moved[i] = P4{o[0], o[1], o[2], src[i].i}; }
        M4f transformation = m4f_identity(), previous = m4f_identity();
        const double dist_threshold = corr_dist_threshold * corr_dist_threshold;
        bool converged = false;
        int nr = 0;
        while (!converged) {
            idx_src.clear(); idx_tgt.clear();
            double TR[16];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += double(transformation.m[i + 4 * k]) * double(guess.m[k + 4 * j]); TR[i + 4 * j] = s; }
            double R[9];
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + 3 * j] = TR[i + 4 * j];
            std::vector<int> nn(N, -1);
            std::vector<float> nd(N, 0.f);
#pragma omp parallel for schedule(dynamic, 256)
            for (long long i = 0; i < (long long)N; ++i) {
                float q[3];
                xform_pt(transformation, moved[size_t(i)].x, moved[size_t(i)].y, moved[size_t(i)].z, q);
                KdTree::Hit h;
                if (tgt_tree.Knn(q, 1, &h) == 1) { nn[size_t(i)] = h.idx; nd[size_t(i)] = h.d2; }
            }
            for (size_t i = 0; i < N; ++i) {
                if (nn[i] < 0 || !(double(nd[i]) < dist_threshold)) continue;
                const double* C1 = &cov_src[9 * i];
                const double* C2 = &cov_tgt[9 * size_t(nn[i])];
                double M1[9], Rt[9], tmp[9];
                mat3_mul(R, C1, M1);
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Rt[r + 3 * c] = R[c + 3 * r];
                mat3_mul(M1, Rt, tmp);
                for (int q = 0; q < 9; ++q) tmp[q] += C2[q];
                inverse3(tmp, &mahal[9 * i]);
                idx_src.push_back(int(i));
                idx_tgt.push_back(nn[i]);
            }
            stats.correspondences = int(idx_src.size());
            previous = transformation;
            if (!estimate(transformation)) { stats.failed = true; break; }
            double delta = 0.0;
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 4; ++l) {
                    const double ratio = (k < 3 && l < 3) ? 1.0 / rotation_epsilon : 1.0 / transformation_epsilon;
                    const double c_delta = ratio * std::fabs(double(previous.m[k + 4 * l]) - double(transformation.m[k + 4 * l]));
                    if (c_delta > delta) delta = c_delta;
                }
            ++nr;
            if (std::getenv("FLO_LOOP_DEBUG"))
                std::fprintf(stderr, "[flo loop] gicp outer %d: corr %d inner_total %d evals %d delta %.9g t = %.9g %.9g %.9g\n", nr, stats.correspondences, stats.inner_total,
                             stats.evaluations, delta, double(transformation.m[12]), double(transformation.m[13]), double(transformation.m[14]));
            if (nr >= max_iterations || delta < 1) { converged = true; previous = transformation; }
        }
        stats.iterations = nr;
        return m4f_mul(previous, guess);  // final_transformation_ = previous_transformation_ * guess
    }
    // Registration::getFitnessScore(): mean squared distance of the transformed source to its nearest target point
    float fitness(const Cloud& src, const M4f& final_t) const {
        double sum = 0.0;
        long long nr = 0;
        for (const P4& p : src) {
            float q[3];
            xform_pt(final_t, p.x, p.y, p.z, q);
            KdTree::Hit h;
            if (tgt_tree.Knn(q, 1, &h) != 1) continue;
            sum += double(h.d2);
            ++nr;
        }
        return nr > 0 ? float(sum / double(nr)) : std::numeric_limits<float>::max();
    }
};

struct LoopStats {
    int32_t ndt_iterations[4], ndt_evaluations[4], ndt_source_points[4], ndt_target_leaves[4];
    int32_t gicp_iterations, gicp_inner_iterations, gicp_evaluations, gicp_correspondences, gicp_source_points, gicp_target_points, gicp_failed, reserved;
    double ndt_score[4];
    double T_after_ndt[16];
};

// LoopClosure::Match (loop_closure.cpp:233-267)
static inline float loop_match(const Cloud& source, const Cloud& target, double* T /*4x4 col-major, in/out*/, LoopStats* st) {
    static const float resolution[4] = {10.0f, 5.0f, 3.0f, 2.0f};
    Ndt ndt;
    ndt.step_size = 0.5;
    ndt.max_iterations = 30;
    for (int s = 0; s < 4; ++s) {
        const float r = resolution[s];
        ndt.resolution = r;
        const Cloud src = voxel_grid(source, r * 0.2f), tgt = voxel_grid(target, r * 0.2f);
        ndt.set_target(tgt);
        const M4f fin = ndt.align(src, m4f_from_d(T));  // pose.cast<float>()
        for (int i = 0; i < 16; ++i) T[i] = double(fin.m[i]);
        if (st) {
            st->ndt_iterations[s] = ndt.stats.iterations; st->ndt_evaluations[s] = ndt.stats.evaluations; st->ndt_source_points[s] = int(src.size());
            st->ndt_target_leaves[s] = int(ndt.cells.searchable.size()); st->ndt_score[s] = ndt.stats.score;
        }
    }
    if (st) for (int i = 0; i < 16; ++i) st->T_after_ndt[i] = T[i];
    const Cloud src = voxel_grid(source, 0.5f), tgt = voxel_grid(target, 0.4f);
    Gicp gicp;
    gicp.max_iterations = 30;
    gicp.corr_dist_threshold = 2.0;
    if (src.size() < size_t(gicp.k_correspondences) || tgt.size() < size_t(gicp.k_correspondences)) return std::numeric_limits<float>::max();
    const M4f fin = gicp.align(src, tgt, m4f_from_d(T));
    for (int i = 0; i < 16; ++i) T[i] = double(fin.m[i]);
    if (st) {
        st->gicp_iterations = gicp.stats.iterations; st->gicp_inner_iterations = gicp.stats.inner_total; st->gicp_evaluations = gicp.stats.evaluations;
        st->gicp_correspondences = gicp.stats.correspondences; st->gicp_source_points = int(src.size()); st->gicp_target_points = int(tgt.size());
        st->gicp_failed = gicp.stats.failed ? 1 : 0; st->reserved = 0;
    }
    return gicp.fitness(src, fin);
}

}  // namespace loop
}  // namespace flo
