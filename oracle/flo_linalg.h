// ============================================================================
// oracle/flo_linalg.h  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Nothing under oracle/ is part of the shipped product.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Restatement of the third-party dense algebra that the reference's
// registration path calls (Eigen 3.3.x, not vendored in /root/reference):
//   * ColPivHouseholderQR<5x3>::solve      loam_point_to_plane_ivox.h:283,
//                                          loam_full_kdtree.h:303,
//                                          loam_point_to_plane_kdtree.h (same)
//   * FullPivHouseholderQR<6x6>::solve     loam_point_to_plane_ivox.h:167,
//                                          loam_full_kdtree.h:141
//   * PartialPivLU<6x6> inverse/determinant icp_optimized.h:129,133,
//                                          incremental_ndt.h:311
//   * Matrix3d::inverse() (cofactors)      incremental_ndt.h:134,151
//   * JacobiSVD<3x3>                       loam_full_kdtree.h:244,
//                                          incremental_ndt.h:166
//   * SO3Hat / SO3Exp                      include/common/math_function.h:52-89
//   * RotationMatrixToRPY                  include/common/math_function.h:139-149
// Eigen itself is absent from this container, so the factorisations follow the
// published Eigen 3.3.7 algorithms from memory ("parity unpinned" for these
// third-party pieces: they are cross-checked against numpy/scipy to 1e-10 in
// tests/test_oracle_linalg.py, and SO3Hat/SO3Exp/RPY against the reference's
// own known-answer tests test/math_function_ut.cpp:9-133,160-192).
// All matrices are column-major like Eigen's default.
// ============================================================================
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <algorithm>

namespace flo {

// ---------------------------------------------------------------------------
// Householder primitives (Eigen/src/Householder/Householder.h)
// ---------------------------------------------------------------------------
// x[0..n-1]; on exit x[1..] holds the essential part.
static inline void make_householder(double* x, int n, double& tau, double& beta) {
    double tail_sq = 0.0;
    for (int i = 1; i < n; ++i) tail_sq += x[i] * x[i];
    const double c0 = x[0];
    const double tol = std::numeric_limits<double>::min();
    if (n == 1 || tail_sq <= tol) {
        tau = 0.0;
        beta = c0;
        for (int i = 1; i < n; ++i) x[i] = 0.0;
    } else {
        beta = std::sqrt(c0 * c0 + tail_sq);
        if (c0 >= 0.0) beta = -beta;
        for (int i = 1; i < n; ++i) x[i] = x[i] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
}

// M: rows x cols block, column-major, leading dimension ld.  ess: rows-1.
static inline void apply_householder_left(double* M, int rows, int cols, int ld,
                                          const double* ess, double tau) {
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

// ---------------------------------------------------------------------------
// ColPivHouseholderQR solve, ROWS x COLS (ROWS >= COLS), least squares.
// Eigen/src/QR/ColPivHouseholderQR.h (3.3.x: LAPACK-style norm down-dating).
// ---------------------------------------------------------------------------
template <int ROWS, int COLS>
static inline void colpiv_qr_solve(const double* A_in /*col-major*/, const double* b, double* x) {
    const double eps = std::numeric_limits<double>::epsilon();
    double qr[ROWS * COLS];
    std::memcpy(qr, A_in, sizeof(qr));
    double hcoef[COLS];
    int perm[COLS];
    double norms_upd[COLS], norms_dir[COLS];
    for (int k = 0; k < COLS; ++k) {
        perm[k] = k;
        double s = 0.0;
        for (int i = 0; i < ROWS; ++i) s += qr[i + k * ROWS] * qr[i + k * ROWS];
        norms_dir[k] = norms_upd[k] = std::sqrt(s);
    }
    double max_norm = norms_upd[0];
    for (int k = 1; k < COLS; ++k) max_norm = std::max(max_norm, norms_upd[k]);
    const double th = max_norm * eps;
    const double threshold_helper = (th * th) / double(ROWS);
    const double norm_downdate_threshold = std::sqrt(eps);
    int nonzero_pivots = COLS;
    for (int k = 0; k < COLS; ++k) {
        int big = k;
        double bigv = norms_upd[k];
        for (int j = k + 1; j < COLS; ++j)
            if (norms_upd[j] > bigv) { bigv = norms_upd[j]; big = j; }
        const double big_sq = bigv * bigv;
        if (nonzero_pivots == COLS && big_sq < threshold_helper * double(ROWS - k)) nonzero_pivots = k;
        if (big != k) {
            for (int i = 0; i < ROWS; ++i) std::swap(qr[i + k * ROWS], qr[i + big * ROWS]);
            std::swap(norms_upd[k], norms_upd[big]);
            std::swap(norms_dir[k], norms_dir[big]);
            std::swap(perm[k], perm[big]);
        }
        double beta;
        make_householder(qr + k + k * ROWS, ROWS - k, hcoef[k], beta);
        qr[k + k * ROWS] = beta;
        if (k + 1 < COLS)
            apply_householder_left(qr + k + (k + 1) * ROWS, ROWS - k, COLS - k - 1, ROWS,
                                   qr + (k + 1) + k * ROWS, hcoef[k]);
        for (int j = k + 1; j < COLS; ++j) {
            if (norms_upd[j] != 0.0) {
                double temp = std::fabs(qr[k + j * ROWS]) / norms_upd[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                const double r = norms_upd[j] / norms_dir[j];
                const double temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    double s = 0.0;
                    for (int i = k + 1; i < ROWS; ++i) s += qr[i + j * ROWS] * qr[i + j * ROWS];
                    norms_dir[j] = std::sqrt(s);
                    norms_upd[j] = norms_dir[j];
                } else {
                    norms_upd[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // solve
    if (nonzero_pivots == 0) { for (int j = 0; j < COLS; ++j) x[j] = 0.0; return; }
    double c[ROWS];
    std::memcpy(c, b, sizeof(c));
    for (int k = 0; k < nonzero_pivots; ++k)
        apply_householder_left(c + k, ROWS - k, 1, ROWS, qr + (k + 1) + k * ROWS, hcoef[k]);
    // upper-triangular back substitution, column-oriented (Eigen triangular_solve_vector)
    for (int i = nonzero_pivots - 1; i >= 0; --i) {
        c[i] /= qr[i + i * ROWS];
        for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r + i * ROWS];
    }
    for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
    for (int i = nonzero_pivots; i < COLS; ++i) x[perm[i]] = 0.0;
}

// ---------------------------------------------------------------------------
// FullPivHouseholderQR<N,N>::solve   (Eigen/src/QR/FullPivHouseholderQR.h)
// ---------------------------------------------------------------------------
template <int N>
static inline void fullpiv_qr_solve(const double* A_in, const double* b, double* x) {
    const double eps = std::numeric_limits<double>::epsilon();
    double qr[N * N];
    std::memcpy(qr, A_in, sizeof(qr));
    double hcoef[N];
    int rows_tr[N], cols_tr[N];
    const double precision = eps * double(N);
    double biggest = 0.0, maxpivot = 0.0;
    int nonzero_pivots = N;
    for (int k = 0; k < N; ++k) {
        // biggest |coeff| of the bottom-right corner, column-major visit, first max wins
        int rb = k, cb = k;
        double bc = std::fabs(qr[k + k * N]);
        for (int j = k; j < N; ++j)
            for (int i = k; i < N; ++i) {
                const double v = std::fabs(qr[i + j * N]);
                if (v > bc) { bc = v; rb = i; cb = j; }
            }
        if (k == 0) biggest = bc;
        if (std::fabs(bc) <= std::fabs(biggest) * precision) {  // isMuchSmallerThan
            nonzero_pivots = k;
            for (int i = k; i < N; ++i) { rows_tr[i] = i; cols_tr[i] = i; hcoef[i] = 0.0; }
            break;
        }
        rows_tr[k] = rb;
        cols_tr[k] = cb;
        if (k != rb)
            for (int j = k; j < N; ++j) std::swap(qr[k + j * N], qr[rb + j * N]);
        if (k != cb)
            for (int i = 0; i < N; ++i) std::swap(qr[i + k * N], qr[i + cb * N]);
        double beta;
        make_householder(qr + k + k * N, N - k, hcoef[k], beta);
        qr[k + k * N] = beta;
        if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
        if (k + 1 < N)
            apply_householder_left(qr + k + (k + 1) * N, N - k, N - k - 1, N,
                                   qr + (k + 1) + k * N, hcoef[k]);
    }
    int perm[N];
    for (int i = 0; i < N; ++i) perm[i] = i;
    for (int k = 0; k < N; ++k) std::swap(perm[k], perm[cols_tr[k]]);
    // rank()
    const double premult = std::fabs(maxpivot) * (eps * double(N));
    int rank = 0;
    for (int i = 0; i < nonzero_pivots; ++i) rank += (std::fabs(qr[i + i * N]) > premult) ? 1 : 0;
    if (rank == 0) { for (int i = 0; i < N; ++i) x[i] = 0.0; return; }
    double c[N];
    std::memcpy(c, b, sizeof(c));
    for (int k = 0; k < rank; ++k) {
        std::swap(c[k], c[rows_tr[k]]);
        apply_householder_left(c + k, N - k, 1, N, qr + (k + 1) + k * N, hcoef[k]);
    }
    for (int i = rank - 1; i >= 0; --i) {
        c[i] /= qr[i + i * N];
        for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r + i * N];
    }
    for (int i = 0; i < rank; ++i) x[perm[i]] = c[i];
    for (int i = rank; i < N; ++i) x[perm[i]] = 0.0;
}

// ---------------------------------------------------------------------------
// PartialPivLU<N,N>: determinant and explicit inverse
// (Eigen/src/LU/PartialPivLU.h, unblocked_lu for size <= 16)
// ---------------------------------------------------------------------------
template <int N>
struct PartialPivLU {
    double lu[N * N];
    int row_tr[N];
    int det_p = 1;
    void compute(const double* A) {
        std::memcpy(lu, A, sizeof(lu));
        int ntr = 0;
        for (int k = 0; k < N; ++k) {
            int rb = k;
            double bc = std::fabs(lu[k + k * N]);
            for (int i = k + 1; i < N; ++i) {
                const double v = std::fabs(lu[i + k * N]);
                if (v > bc) { bc = v; rb = i; }
            }
            row_tr[k] = rb;
            if (bc != 0.0) {
                if (k != rb) {
                    for (int j = 0; j < N; ++j) std::swap(lu[k + j * N], lu[rb + j * N]);
                    ++ntr;
                }
                for (int i = k + 1; i < N; ++i) lu[i + k * N] /= lu[k + k * N];
            }
            for (int j = k + 1; j < N; ++j)
                for (int i = k + 1; i < N; ++i) lu[i + j * N] -= lu[i + k * N] * lu[k + j * N];
        }
        det_p = (ntr % 2) ? -1 : 1;
    }
    double determinant() const {
        double p = lu[0];
        for (int i = 1; i < N; ++i) p *= lu[i + i * N];
        return double(det_p) * p;
    }
    void inverse(double* inv) const {
        // dst = P * I
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) inv[i + j * N] = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < N; ++k)
            if (row_tr[k] != k)
                for (int j = 0; j < N; ++j) std::swap(inv[k + j * N], inv[row_tr[k] + j * N]);
        for (int j = 0; j < N; ++j) {
            double* c = inv + j * N;
            // unit-lower forward substitution (column oriented)
            for (int i = 0; i < N; ++i)
                for (int r = i + 1; r < N; ++r) c[r] -= c[i] * lu[r + i * N];
            // upper back substitution
            for (int i = N - 1; i >= 0; --i) {
                c[i] /= lu[i + i * N];
                for (int r = 0; r < i; ++r) c[r] -= c[i] * lu[r + i * N];
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Matrix3d::inverse()  (Eigen/src/LU/InverseImpl.h compute_inverse_size3)
// ---------------------------------------------------------------------------
static inline double cofactor3(const double* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 + j1 * 3] * m[i2 + j2 * 3] - m[i1 + j2 * 3] * m[i2 + j1 * 3];
}
static inline void inverse3(const double* m, double* inv) {
    const double c00 = cofactor3(m, 0, 0), c10 = cofactor3(m, 1, 0), c20 = cofactor3(m, 2, 0);
    const double det = (c00 * m[0] + c10 * m[1]) + c20 * m[2];
    const double invdet = 1.0 / det;
    inv[0 + 0 * 3] = c00 * invdet;
    inv[0 + 1 * 3] = c10 * invdet;
    inv[0 + 2 * 3] = c20 * invdet;
    inv[1 + 0 * 3] = cofactor3(m, 0, 1) * invdet;
    inv[1 + 1 * 3] = cofactor3(m, 1, 1) * invdet;
    inv[1 + 2 * 3] = cofactor3(m, 2, 1) * invdet;
    inv[2 + 0 * 3] = cofactor3(m, 0, 2) * invdet;
    inv[2 + 1 * 3] = cofactor3(m, 1, 2) * invdet;
    inv[2 + 2 * 3] = cofactor3(m, 2, 2) * invdet;
}

// ---------------------------------------------------------------------------
// JacobiSVD 3x3, full U and V (Eigen/src/SVD/JacobiSVD.h, two-sided Jacobi)
// ---------------------------------------------------------------------------
struct JRot { double c, s; };
static inline bool make_jacobi(double x, double y, double z, JRot& r) {
    const double deno = 2.0 * std::fabs(y);
    if (deno < std::numeric_limits<double>::min()) { r.c = 1.0; r.s = 0.0; return false; }
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    const double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    r.c = n;
    return true;
}
// rows p,q of W:  x <- c x + s y ;  y <- -s x + c y
static inline void rot_left(double* W, int p, int q, JRot j) {
    for (int k = 0; k < 3; ++k) {
        const double xi = W[p + k * 3], yi = W[q + k * 3];
        W[p + k * 3] = j.c * xi + j.s * yi;
        W[q + k * 3] = -j.s * xi + j.c * yi;
    }
}
// cols p,q of W with j.transpose():  x <- c x - s y ; y <- s x + c y
static inline void rot_right(double* W, int p, int q, JRot j) {
    for (int k = 0; k < 3; ++k) {
        const double xi = W[k + p * 3], yi = W[k + q * 3];
        W[k + p * 3] = j.c * xi - j.s * yi;
        W[k + q * 3] = j.s * xi + j.c * yi;
    }
}
static inline void jacobi_svd3(const double* A, double* U, double* S, double* V) {
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    const double consider_zero = std::numeric_limits<double>::min();
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(A[i]));
    if (scale == 0.0) scale = 1.0;
    double W[9];
    for (int i = 0; i < 9; ++i) { W[i] = A[i] / scale; U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    double max_diag = std::max(std::fabs(W[0]), std::max(std::fabs(W[4]), std::fabs(W[8])));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 200) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = std::max(consider_zero, precision * max_diag);
                if (std::fabs(W[p + q * 3]) > threshold || std::fabs(W[q + p * 3]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd
                    double m00 = W[p + p * 3], m01 = W[p + q * 3], m10 = W[q + p * 3], m11 = W[q + q * 3];
                    JRot rot1;
                    const double t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < std::numeric_limits<double>::min()) { rot1.s = 0.0; rot1.c = 1.0; }
                    else {
                        const double u = t / d, tmp = std::sqrt(1.0 + u * u);
                        rot1.s = 1.0 / tmp; rot1.c = u / tmp;
                    }
                    // m.applyOnTheLeft(0,1,rot1)
                    const double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
                    const double n11 = -rot1.s * m01 + rot1.c * m11;
                    JRot jr;
                    make_jacobi(n00, n01, n11, jr);
                    // j_left = rot1 * jr.transpose()
                    JRot jrt{jr.c, -jr.s};
                    JRot jl{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
                    rot_left(W, p, q, jl);
                    JRot jlt{jl.c, -jl.s};
                    // U.applyOnTheRight(p,q,j_left.transpose()) -> uses (j_left^T)^T = j_left
                    {   // apply_rotation_in_the_plane(col p, col q, jlt.transpose() = jl)
                        for (int k = 0; k < 3; ++k) {
                            const double xi = U[k + p * 3], yi = U[k + q * 3];
                            U[k + p * 3] = jl.c * xi + jl.s * yi;
                            U[k + q * 3] = -jl.s * xi + jl.c * yi;
                        }
                        (void)jlt;
                    }
                    rot_right(W, p, q, jr);
                    rot_right(V, p, q, jr);
                    max_diag = std::max(max_diag, std::max(std::fabs(W[p + p * 3]), std::fabs(W[q + q * 3])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W[i + i * 3];
        S[i] = std::fabs(a);
        if (a < 0.0) for (int k = 0; k < 3; ++k) U[k + i * 3] = -U[k + i * 3];
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        double mx = S[i];
        for (int k = i + 1; k < 3; ++k) if (S[k] > mx) { mx = S[k]; pos = k; }
        if (mx == 0.0) break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int k = 0; k < 3; ++k) { std::swap(U[k + i * 3], U[k + pos * 3]); std::swap(V[k + i * 3], V[k + pos * 3]); }
        }
    }
}

// ---------------------------------------------------------------------------
// SO3Hat / SO3Exp / RotationMatrixToRPY  (include/common/math_function.h)
// ---------------------------------------------------------------------------
static inline void so3_hat(const double v[3], double* M /*3x3 col-major*/) {  // :52-64
    for (int i = 0; i < 9; ++i) M[i] = 0.0;
    M[0 + 1 * 3] = -v[2];
    M[0 + 2 * 3] = +v[1];
    M[1 + 2 * 3] = -v[0];
    M[1 + 0 * 3] = +v[2];
    M[2 + 0 * 3] = -v[1];
    M[2 + 1 * 3] = +v[0];
}
static inline void so3_exp(const double v[3], double* R /*3x3 col-major*/) {  // :74-89
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    const double sq = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double theta = std::sqrt(sq);
    if (theta > std::numeric_limits<double>::epsilon()) {
        double a[3] = {v[0], v[1], v[2]};
        if (sq > 0.0) { const double n = std::sqrt(sq); a[0] = v[0] / n; a[1] = v[1] / n; a[2] = v[2] / n; }
        const double c = std::cos(theta), s = std::sin(theta);
        double hat[9];
        so3_hat(a, hat);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const double id = (i == j) ? 1.0 : 0.0;
                R[i + j * 3] = (c * id + ((1.0 - c) * a[i]) * a[j]) + s * hat[i + j * 3];
            }
    }
}
static inline void rotation_to_rpy(const double* R, double rpy[3]) {  // :139-149
    rpy[0] = std::atan2(R[2 + 1 * 3], R[2 + 2 * 3]);
    rpy[1] = std::asin(-R[2 + 0 * 3]);
    rpy[2] = std::atan2(R[1 + 0 * 3], R[0 + 0 * 3]);
}
// C = A * B, 3x3 col-major, Eigen lazy-product coefficient order
static inline void mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            T[i + j * 3] = (A[i + 0 * 3] * B[0 + j * 3] + A[i + 1 * 3] * B[1 + j * 3]) + A[i + 2 * 3] * B[2 + j * 3];
    std::memcpy(C, T, sizeof(T));
}
static inline double norm3(const double* v) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

}  // namespace flo
