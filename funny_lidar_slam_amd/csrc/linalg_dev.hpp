// linalg_dev.hpp -- small dense FP64 factorisations for gfx950 device code.
//
// Two families:
//  * per-lane, register-resident (static indexing only, no scratch):
//      plane_fit_5x3      5x3 least squares  A x = -1  by column-pivoted Householder QR
//                         (what Eigen's colPivHouseholderQr().solve does at
//                          loam_point_to_plane_ivox.h:283, loam_full_kdtree.h:303)
//      sym_eig_svd3       singular values / V of a symmetric 3x3 by two-sided Jacobi
//                         (JacobiSVD at loam_full_kdtree.h:244)
//  * helpers of the Gauss-Newton tail (one lane, once per iteration):
//      so3_exp_dev        include/common/math_function.h:74-89
//      mat3_mul_dev, norm3d
//    (the 6x6 solvers -- fullPivHouseholderQr().solve, PartialPivLU inverse -- are wave-cooperative and live
//     in wave_solve.hpp)
// Compiled with -ffp-contract=off: no FMA contraction, so the arithmetic is the
// same sequence of IEEE mul/add the CPU path performs.
#pragma once
#include <hip/hip_runtime.h>

namespace fls {

#define FLS_DBL_MIN 2.2250738585072014e-308
#define FLS_DBL_EPS 2.220446049250313e-16

// ---------------------------------------------------------------------------------------------
// register-resident 5x3 column-pivoted Householder QR least squares, rhs = -1
// q[c][r]: column c, row r.  All loops fully unrolled -> static register indexing.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }
__device__ __forceinline__ void swap_i(int& a, int& b) { const int t = a; a = b; b = t; }

__device__ __forceinline__ void plane_fit_5x3(const double (&A)[3][5], double (&x)[3]) {
    double q[3][5];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 5; ++r) q[c][r] = A[c][r];
    double nu[3], nd[3], hc[3];
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 5; ++r) s += q[c][r] * q[c][r];
        nu[c] = nd[c] = sqrt(s);
    }
    double mx = nu[0];
    mx = nu[1] > mx ? nu[1] : mx;
    mx = nu[2] > mx ? nu[2] : mx;
    const double th = mx * FLS_DBL_EPS;
    const double threshold_helper = (th * th) / 5.0;
    const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
    int nzp = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k;
        double bv = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (nu[j] > bv) { bv = nu[j]; big = j; }
        if (nzp == 3 && bv * bv < threshold_helper * double(5 - k)) nzp = k;
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (big == j) {
#pragma unroll
                for (int r = 0; r < 5; ++r) swap_d(q[k][r], q[j][r]);
                swap_d(nu[k], nu[j]);
                swap_d(nd[k], nd[j]);
                swap_i(perm[k], perm[j]);
            }
        // Householder on column k, rows k..4
        double tail = 0.0;
#pragma unroll
        for (int r = k + 1; r < 5; ++r) tail += q[k][r] * q[k][r];
        const double c0 = q[k][k];
        double tau, beta;
        if (tail <= FLS_DBL_MIN) {
            tau = 0.0;
            beta = c0;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) q[k][r] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) q[k][r] = q[k][r] / den;
            tau = (beta - c0) / beta;
        }
        q[k][k] = beta;
        hc[k] = tau;
        // apply to the trailing columns
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (tau != 0.0) {
                double tmp = 0.0;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) tmp += q[k][r] * q[j][r];
                tmp += q[j][k];
                q[j][k] -= tau * tmp;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) q[j][r] -= (tau * q[k][r]) * tmp;
            }
        }
        // LAPACK-style column-norm down-dating
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0.0) {
                double temp = fabs(q[j][k]) / nu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                const double rr = nu[j] / nd[j];
                const double temp2 = temp * (rr * rr);
                if (temp2 <= downdate_thr) {
                    double s = 0.0;
#pragma unroll
                    for (int r = k + 1; r < 5; ++r) s += q[j][r] * q[j][r];
                    nd[j] = sqrt(s);
                    nu[j] = nd[j];
                } else {
                    nu[j] *= sqrt(temp);
                }
            }
        }
    }
    // c = Q^T b, b = -1
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nzp && hc[k] != 0.0) {
            double tmp = 0.0;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) tmp += q[k][r] * c[r];
            tmp += c[k];
            c[k] -= hc[k] * tmp;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) c[r] -= (hc[k] * q[k][r]) * tmp;
        }
    }
    // back substitution on the leading nzp x nzp triangle (column oriented)
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nzp) {
            c[i] /= q[i][i];
#pragma unroll
            for (int r = 0; r < i; ++r) c[r] -= c[i] * q[i][r];
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) x[j] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (perm[i] == j) x[j] = (i < nzp) ? c[i] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Two-sided Jacobi SVD of a 3x3 (Eigen JacobiSVD algorithm), registers only.
// W,V indexed [col][row].  Returns singular values sorted descending and V.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_pair(double (&W)[3][3], double (&V)[3][3], const int p, const int q,
                                            double& max_diag, bool& finished) {
    const double precision = 2.0 * FLS_DBL_EPS;
    double threshold = precision * max_diag;
    threshold = threshold > FLS_DBL_MIN ? threshold : FLS_DBL_MIN;
    if (fabs(W[q][p]) > threshold || fabs(W[p][q]) > threshold) {
        finished = false;
        const double m00 = W[p][p], m01 = W[q][p], m10 = W[p][q], m11 = W[q][q];
        double r1c, r1s;
        const double t = m00 + m11, d = m10 - m01;
        if (fabs(d) < FLS_DBL_MIN) { r1s = 0.0; r1c = 1.0; }
        else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
        const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
        double jc, js;
        const double deno = 2.0 * fabs(n01);
        if (deno < FLS_DBL_MIN) { jc = 1.0; js = 0.0; }
        else {
            const double tau = (n00 - n11) / deno;
            const double w = sqrt(tau * tau + 1.0);
            const double tt = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
            const double sign_t = tt > 0.0 ? 1.0 : -1.0;
            const double n = 1.0 / sqrt(tt * tt + 1.0);
            js = -sign_t * (n01 / fabs(n01)) * fabs(tt) * n;
            jc = n;
        }
        // j_left = rot1 * j_right^T
        const double lc = r1c * jc - r1s * (-js), ls = r1c * (-js) + r1s * jc;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // rows p,q
            const double xi = W[k][p], yi = W[k][q];
            W[k][p] = lc * xi + ls * yi;
            W[k][q] = -ls * xi + lc * yi;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // cols p,q of W and V with j_right
            const double xi = W[p][k], yi = W[q][k];
            W[p][k] = jc * xi - js * yi;
            W[q][k] = js * xi + jc * yi;
            const double vx = V[p][k], vy = V[q][k];
            V[p][k] = jc * vx - js * vy;
            V[q][k] = js * vx + jc * vy;
        }
        double a = fabs(W[p][p]), b = fabs(W[q][q]);
        a = a > b ? a : b;
        max_diag = max_diag > a ? max_diag : a;
    }
}

__device__ __forceinline__ void jacobi_svd3_v(const double (&A)[3][3], double (&S)[3], double (&V)[3][3]) {
    double scale = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double a = fabs(A[c][r]); scale = a > scale ? a : scale; }
    if (scale == 0.0) scale = 1.0;
    double W[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) { W[c][r] = A[c][r] / scale; V[c][r] = (c == r) ? 1.0 : 0.0; }
    double max_diag = fabs(W[0][0]);
    max_diag = fabs(W[1][1]) > max_diag ? fabs(W[1][1]) : max_diag;
    max_diag = fabs(W[2][2]) > max_diag ? fabs(W[2][2]) : max_diag;
    bool finished = false;
    for (int sweep = 0; sweep < 64 && !finished; ++sweep) {
        finished = true;
        jacobi_pair(W, V, 1, 0, max_diag, finished);
        jacobi_pair(W, V, 2, 0, max_diag, finished);
        jacobi_pair(W, V, 2, 1, max_diag, finished);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) S[i] = fabs(W[i][i]) * scale;
    // sort descending (selection, first max wins), swapping V columns
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pos = i;
        double mxv = S[i];
#pragma unroll
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > mxv) { mxv = S[k]; pos = k; }
#pragma unroll
        for (int k = i + 1; k < 3; ++k)
            if (pos == k && mxv != 0.0) {
                swap_d(S[i], S[k]);
#pragma unroll
                for (int r = 0; r < 3; ++r) swap_d(V[i][r], V[k][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// small helpers of the Gauss-Newton tail (the 6x6 solvers themselves are wave-cooperative: wave_solve.hpp)
// ---------------------------------------------------------------------------------------------
__device__ inline double norm3d(const double* v) { return sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

// SO3Exp (math_function.h:74-89): identity if |v| <= eps, else Rodrigues.  R 3x3 column-major.
#ifndef FLS_TAIL_SINCOS
#define FLS_TAIL_SINCOS 1  // 0: separate cos() and sin() calls (A/B builds)
#endif
__device__ inline void so3_exp_dev(const double* v, double* R) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    const double sq = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double theta = sqrt(sq);
    if (theta > FLS_DBL_EPS) {
        const double a[3] = {v[0] / theta, v[1] / theta, v[2] / theta};
#if FLS_TAIL_SINCOS
        double s, c;
        sincos(theta, &s, &c);  // one argument reduction for both (the tail is one lone wave: every instruction costs ~8 cycles)
        // R(i, j) = (c [i == j] + ((1 - c) a_i) a_j) + s hat(a)(i, j) with the products by 1.0 and 0.0 and the additions of 0.0 left out: the same
        // bits for finite theta (x * 1.0 == x, x + 0.0 == x; only the sign of an exact zero can differ), 24 instructions instead of ~95
        const double w[3] = {(1.0 - c) * a[0], (1.0 - c) * a[1], (1.0 - c) * a[2]};
        const double sa[3] = {s * a[0], s * a[1], s * a[2]};
        R[0] = c + w[0] * a[0];  R[4] = c + w[1] * a[1];  R[8] = c + w[2] * a[2];
        R[1] = w[1] * a[0] + sa[2];  R[2] = w[2] * a[0] - sa[1];   // column 0: hat = (0, a2, -a1)
        R[3] = w[0] * a[1] - sa[2];  R[5] = w[2] * a[1] + sa[0];   // column 1: hat = (-a2, 0, a0)
        R[6] = w[0] * a[2] + sa[1];  R[7] = w[1] * a[2] - sa[0];   // column 2: hat = (a1, -a0, 0)
#else
        const double c = cos(theta), s = sin(theta);
        double hat[9] = {0.0, a[2], -a[1], -a[2], 0.0, a[0], a[1], -a[0], 0.0};  // column-major SO3Hat(a)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const double id = (i == j) ? 1.0 : 0.0;
                R[i + j * 3] = (c * id + ((1.0 - c) * a[i]) * a[j]) + s * hat[i + j * 3];
            }
#endif
    }
}
// C = A*B, 3x3 column-major
__device__ inline void mat3_mul_dev(const double* A, const double* B, double* C) {
    double T[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            T[i + j * 3] = (A[i + 0 * 3] * B[0 + j * 3] + A[i + 1 * 3] * B[1 + j * 3]) + A[i + 2 * 3] * B[2 + j * 3];
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}

}  // namespace fls
