// linalg_dev.hpp -- small dense FP64 factorisations for gfx950 device code.
//
// Two families:
//  * per-lane, register-resident (static indexing only, no scratch):
//      plane_fit_5x3      5x3 least squares  A x = -1  by column-pivoted Householder QR
//                         (what Eigen's colPivHouseholderQr().solve does at
//                          loam_point_to_plane_ivox.h:283, loam_full_kdtree.h:303)
//      sym_eig_svd3       singular values / V of a symmetric 3x3 by two-sided Jacobi
//                         (JacobiSVD at loam_full_kdtree.h:244)
//  * one-thread, LDS/global-resident (dynamic indexing allowed), used once per
//    Gauss-Newton iteration by the solve kernel:
//      fullpiv_qr_solve6  H.fullPivHouseholderQr().solve(g)   loam_point_to_plane_ivox.h:167
//      lu6_inverse_det    H.determinant(), H.inverse()        icp_optimized.h:129,133, incremental_ndt.h:311
//      so3_exp            include/common/math_function.h:74-89
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
// one-thread solvers on memory-resident (LDS) column-major 6x6
// ---------------------------------------------------------------------------------------------
__device__ inline void householder_make(double* x, int n, double& tau, double& beta) {
    double tail = 0.0;
    for (int i = 1; i < n; ++i) tail += x[i] * x[i];
    const double c0 = x[0];
    if (n == 1 || tail <= FLS_DBL_MIN) {
        tau = 0.0;
        beta = c0;
        for (int i = 1; i < n; ++i) x[i] = 0.0;
    } else {
        beta = sqrt(c0 * c0 + tail);
        if (c0 >= 0.0) beta = -beta;
        const double den = c0 - beta;
        for (int i = 1; i < n; ++i) x[i] = x[i] / den;
        tau = (beta - c0) / beta;
    }
}
__device__ inline void householder_apply_left(double* M, int rows, int cols, int ld, const double* ess, double tau) {
    if (rows == 1) {
        for (int j = 0; j < cols; ++j) M[j * ld] *= (1.0 - tau);
        return;
    }
    if (tau == 0.0) return;
    for (int j = 0; j < cols; ++j) {
        double* col = M + j * ld;
        double tmp = 0.0;
        for (int i = 1; i < rows; ++i) tmp += ess[i - 1] * col[i];
        tmp += col[0];
        col[0] -= tau * tmp;
        for (int i = 1; i < rows; ++i) col[i] -= (tau * ess[i - 1]) * tmp;
    }
}

// qr: 36 doubles work space holding H on entry (destroyed); c: 6 doubles holding g on entry; x out.
__device__ inline void fullpiv_qr_solve6(double* qr, double* c, double* x, double* hcoef, int* rows_tr, int* perm) {
    const int N = 6;
    const double precision = FLS_DBL_EPS * 6.0;
    double biggest = 0.0, maxpivot = 0.0;
    int nonzero_pivots = N;
    for (int i = 0; i < N; ++i) perm[i] = i;
    int cols_tr[6];
    for (int k = 0; k < N; ++k) {
        int rb = k, cb = k;
        double bc = fabs(qr[k + k * N]);
        for (int j = k; j < N; ++j)
            for (int i = k; i < N; ++i) {
                const double v = fabs(qr[i + j * N]);
                if (v > bc) { bc = v; rb = i; cb = j; }
            }
        if (k == 0) biggest = bc;
        if (fabs(bc) <= fabs(biggest) * precision) {
            nonzero_pivots = k;
            for (int i = k; i < N; ++i) { rows_tr[i] = i; cols_tr[i] = i; hcoef[i] = 0.0; }
            break;
        }
        rows_tr[k] = rb;
        cols_tr[k] = cb;
        if (k != rb)
            for (int j = k; j < N; ++j) { const double t = qr[k + j * N]; qr[k + j * N] = qr[rb + j * N]; qr[rb + j * N] = t; }
        if (k != cb)
            for (int i = 0; i < N; ++i) { const double t = qr[i + k * N]; qr[i + k * N] = qr[i + cb * N]; qr[i + cb * N] = t; }
        double beta;
        householder_make(qr + k + k * N, N - k, hcoef[k], beta);
        qr[k + k * N] = beta;
        if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
        if (k + 1 < N) householder_apply_left(qr + k + (k + 1) * N, N - k, N - k - 1, N, qr + (k + 1) + k * N, hcoef[k]);
    }
    for (int k = 0; k < N; ++k) { const int t = perm[k]; perm[k] = perm[cols_tr[k]]; perm[cols_tr[k]] = t; }
    const double premult = fabs(maxpivot) * (FLS_DBL_EPS * 6.0);
    int rank = 0;
    for (int i = 0; i < nonzero_pivots; ++i) rank += (fabs(qr[i + i * N]) > premult) ? 1 : 0;
    for (int i = 0; i < N; ++i) x[i] = 0.0;
    if (rank == 0) return;
    for (int k = 0; k < rank; ++k) {
        const double t = c[k]; c[k] = c[rows_tr[k]]; c[rows_tr[k]] = t;
        householder_apply_left(c + k, N - k, 1, N, qr + (k + 1) + k * N, hcoef[k]);
    }
    for (int i = rank - 1; i >= 0; --i) {
        c[i] /= qr[i + i * N];
        for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r + i * N];
    }
    for (int i = 0; i < rank; ++i) x[perm[i]] = c[i];
}

// lu: 36 doubles holding H on entry; inv: 36 doubles out; returns determinant.
__device__ inline double lu6_inverse_det(double* lu, double* inv, int* row_tr) {
    const int N = 6;
    int ntr = 0;
    for (int k = 0; k < N; ++k) {
        int rb = k;
        double bc = fabs(lu[k + k * N]);
        for (int i = k + 1; i < N; ++i) {
            const double v = fabs(lu[i + k * N]);
            if (v > bc) { bc = v; rb = i; }
        }
        row_tr[k] = rb;
        if (bc != 0.0) {
            if (k != rb) {
                for (int j = 0; j < N; ++j) { const double t = lu[k + j * N]; lu[k + j * N] = lu[rb + j * N]; lu[rb + j * N] = t; }
                ++ntr;
            }
            for (int i = k + 1; i < N; ++i) lu[i + k * N] /= lu[k + k * N];
        }
        for (int j = k + 1; j < N; ++j)
            for (int i = k + 1; i < N; ++i) lu[i + j * N] -= lu[i + k * N] * lu[k + j * N];
    }
    double det = lu[0];
    for (int i = 1; i < N; ++i) det *= lu[i + i * N];
    det = (ntr & 1) ? -det : det;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) inv[i + j * N] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < N; ++k)
        if (row_tr[k] != k)
            for (int j = 0; j < N; ++j) { const double t = inv[k + j * N]; inv[k + j * N] = inv[row_tr[k] + j * N]; inv[row_tr[k] + j * N] = t; }
    for (int j = 0; j < N; ++j) {
        double* c = inv + j * N;
        for (int i = 0; i < N; ++i)
            for (int r = i + 1; r < N; ++r) c[r] -= c[i] * lu[r + i * N];
        for (int i = N - 1; i >= 0; --i) {
            c[i] /= lu[i + i * N];
            for (int r = 0; r < i; ++r) c[r] -= c[i] * lu[r + i * N];
        }
    }
    return det;
}

__device__ inline double norm3d(const double* v) { return sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

// SO3Exp (math_function.h:74-89): identity if |v| <= eps, else Rodrigues.  R 3x3 column-major.
__device__ inline void so3_exp_dev(const double* v, double* R) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    const double sq = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double theta = sqrt(sq);
    if (theta > FLS_DBL_EPS) {
        const double a[3] = {v[0] / theta, v[1] / theta, v[2] / theta};
        const double c = cos(theta), s = sin(theta);
        double hat[9] = {0.0, a[2], -a[1], -a[2], 0.0, a[0], a[1], -a[0], 0.0};  // column-major SO3Hat(a)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const double id = (i == j) ? 1.0 : 0.0;
                R[i + j * 3] = (c * id + ((1.0 - c) * a[i]) * a[j]) + s * hat[i + j * 3];
            }
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
