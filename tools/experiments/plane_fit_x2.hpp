// plane_fit_x2.hpp -- branch-free restatement of the 5x3 column-pivoted Householder plane fit and of the point-to-plane
// residual (linalg_dev.hpp::plane_fit_5x3, kernels_p2plane.hpp::plane_residual_dev), generic over the "real" type:
//     R = double : one point per lane
//     R = D2     : TWO points per lane; every operation is applied to both, back to back, so the two dependent FP64
//                  chains interleave in the instruction stream (software pipelining by construction: the fit phase runs
//                  one wave per SIMD, there is no other wave to hide an FP64 result latency behind)
// Every data-dependent branch of the original is a select here: both sides are evaluated with the SAME IEEE operations in
// the SAME order, the untaken side is discarded (NaN / Inf produced by an untaken side never reaches a result), so each
// point's numbers are bit-identical to the branchy version (checked on the CPU: tests/host/device_math_test.cpp runs
// this header as host code against the oracle's colpiv_qr_solve<5,3> and plane_residual).
// Arithmetic follows Eigen 3.3 ColPivHouseholderQR<5x3>::solve as restated in oracle/flo_linalg.h:76-149
// (reference call sites: loam_point_to_plane_ivox.h:283, loam_full_kdtree.h:303, loam_point_to_plane_kdtree.h:231).
#pragma once
#include <hip/hip_runtime.h>
#include "linalg_dev.hpp"

namespace fls {

#define FLS_HD __host__ __device__ __forceinline__

struct D2 { double a, b; };
struct M2 { bool a, b; };

// ---- the operations the algorithm needs, for double and for D2 --------------------------------------------------
FLS_HD D2 operator+(const D2 x, const D2 y) { return D2{x.a + y.a, x.b + y.b}; }
FLS_HD D2 operator-(const D2 x, const D2 y) { return D2{x.a - y.a, x.b - y.b}; }
FLS_HD D2 operator*(const D2 x, const D2 y) { return D2{x.a * y.a, x.b * y.b}; }
FLS_HD D2 operator/(const D2 x, const D2 y) { return D2{x.a / y.a, x.b / y.b}; }
FLS_HD D2 operator-(const D2 x) { return D2{-x.a, -x.b}; }
FLS_HD M2 operator<(const D2 x, const D2 y) { return M2{x.a < y.a, x.b < y.b}; }
FLS_HD M2 operator<=(const D2 x, const D2 y) { return M2{x.a <= y.a, x.b <= y.b}; }
FLS_HD M2 operator>(const D2 x, const D2 y) { return M2{x.a > y.a, x.b > y.b}; }
FLS_HD M2 operator>=(const D2 x, const D2 y) { return M2{x.a >= y.a, x.b >= y.b}; }
FLS_HD M2 operator==(const D2 x, const D2 y) { return M2{x.a == y.a, x.b == y.b}; }
FLS_HD M2 operator!=(const D2 x, const D2 y) { return M2{x.a != y.a, x.b != y.b}; }
FLS_HD M2 operator&&(const M2 x, const M2 y) { return M2{x.a && y.a, x.b && y.b}; }
FLS_HD M2 operator!(const M2 x) { return M2{!x.a, !x.b}; }

template <class R> struct RealTraits;
template <> struct RealTraits<double> {
    using Mask = bool;
    static FLS_HD double splat(const double v) { return v; }
    static FLS_HD bool mtrue() { return true; }
};
template <> struct RealTraits<D2> {
    using Mask = M2;
    static FLS_HD D2 splat(const double v) { return D2{v, v}; }
    static FLS_HD M2 mtrue() { return M2{true, true}; }
};
FLS_HD double rsel(const bool m, const double x, const double y) { return m ? x : y; }
FLS_HD D2 rsel(const M2 m, const D2 x, const D2 y) { return D2{m.a ? x.a : y.a, m.b ? x.b : y.b}; }
FLS_HD double rsqrt_(const double x) { return sqrt(x); }
FLS_HD D2 rsqrt_(const D2 x) { return D2{sqrt(x.a), sqrt(x.b)}; }
FLS_HD double rabs_(const double x) { return fabs(x); }
FLS_HD D2 rabs_(const D2 x) { return D2{fabs(x.a), fabs(x.b)}; }
template <class R, class M> FLS_HD void swap_if(const M m, R& x, R& y) {
    const R nx = rsel(m, y, x), ny = rsel(m, x, y);
    x = nx; y = ny;
}

// ---------------------------------------------------------------------------------------------------------------
// 5x3 least squares  A x = -1  (A[c][r]: column c, row r)
// ---------------------------------------------------------------------------------------------------------------
template <class R>
FLS_HD void plane_fit_5x3_bf(const R (&A)[3][5], R (&x)[3]) {
    using TR = RealTraits<R>;
    using M = typename TR::Mask;
    const R zero = TR::splat(0.0), one = TR::splat(1.0);
    R q[3][5];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 5; ++r) q[c][r] = A[c][r];
    R nu[3], nd[3], hc[3];
    R pid[3] = {TR::splat(0.0), TR::splat(1.0), TR::splat(2.0)};  // original column of each position (the permutation)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        R s = zero;
#pragma unroll
        for (int r = 0; r < 5; ++r) s = s + q[c][r] * q[c][r];
        nu[c] = nd[c] = rsqrt_(s);
    }
    R mx = nu[0];
    mx = rsel(nu[1] > mx, nu[1], mx);
    mx = rsel(nu[2] > mx, nu[2], mx);
    const R th = mx * TR::splat(FLS_DBL_EPS);
    const R threshold_helper = (th * th) / TR::splat(5.0);
    const R downdate_thr = TR::splat(1.4901161193847656e-08);  // sqrt(eps)
    R nzp = TR::splat(3.0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // pivot column: first maximum of nu[k..2]
        R bv = nu[k];
        M gt[3] = {!TR::mtrue(), !TR::mtrue(), !TR::mtrue()};
#pragma unroll
        for (int j = k + 1; j < 3; ++j) { gt[j] = nu[j] > bv; bv = rsel(gt[j], nu[j], bv); }
        nzp = rsel((nzp == TR::splat(3.0)) && (bv * bv < threshold_helper * TR::splat(double(5 - k))), TR::splat(double(k)), nzp);
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            M is_big = gt[j];  // big == j: j won and no later column beat it
#pragma unroll
            for (int j2 = j + 1; j2 < 3; ++j2) is_big = is_big && !gt[j2];
#pragma unroll
            for (int r = 0; r < 5; ++r) swap_if(is_big, q[k][r], q[j][r]);
            swap_if(is_big, nu[k], nu[j]);
            swap_if(is_big, nd[k], nd[j]);
            swap_if(is_big, pid[k], pid[j]);
        }
        // Householder on column k, rows k..4
        R tail = zero;
#pragma unroll
        for (int r = k + 1; r < 5; ++r) tail = tail + q[k][r] * q[k][r];
        const R c0 = q[k][k];
        const M triv = tail <= TR::splat(FLS_DBL_MIN);
        R beta_n = rsqrt_(c0 * c0 + tail);
        beta_n = rsel(c0 >= zero, -beta_n, beta_n);
        const R den = c0 - beta_n;
        const R tau = rsel(triv, zero, (beta_n - c0) / beta_n);
        const R beta = rsel(triv, c0, beta_n);
#pragma unroll
        for (int r = k + 1; r < 5; ++r) q[k][r] = rsel(triv, zero, q[k][r] / den);
        q[k][k] = beta;
        hc[k] = tau;
        const M app = tau != zero;
        // apply to the trailing columns
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            R tmp = zero;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) tmp = tmp + q[k][r] * q[j][r];
            tmp = tmp + q[j][k];
            q[j][k] = rsel(app, q[j][k] - tau * tmp, q[j][k]);
#pragma unroll
            for (int r = k + 1; r < 5; ++r) q[j][r] = rsel(app, q[j][r] - (tau * q[k][r]) * tmp, q[j][r]);
        }
        // LAPACK-style column-norm down-dating
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            const M nz = nu[j] != zero;
            R temp = rabs_(q[j][k]) / nu[j];
            temp = (one + temp) * (one - temp);
            temp = rsel(temp < zero, zero, temp);
            const R rr = nu[j] / nd[j];
            const R temp2 = temp * (rr * rr);
            const M redo = temp2 <= downdate_thr;
            R s = zero;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) s = s + q[j][r] * q[j][r];
            const R fresh = rsqrt_(s);
            const R scaled = nu[j] * rsqrt_(temp);
            nd[j] = rsel(nz && redo, fresh, nd[j]);
            nu[j] = rsel(nz, rsel(redo, fresh, scaled), nu[j]);
        }
    }
    // c = Q^T b, b = -1
    R c[5] = {TR::splat(-1.0), TR::splat(-1.0), TR::splat(-1.0), TR::splat(-1.0), TR::splat(-1.0)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const M app = (TR::splat(double(k)) < nzp) && (hc[k] != zero);
        R tmp = zero;
#pragma unroll
        for (int r = k + 1; r < 5; ++r) tmp = tmp + q[k][r] * c[r];
        tmp = tmp + c[k];
        c[k] = rsel(app, c[k] - hc[k] * tmp, c[k]);
#pragma unroll
        for (int r = k + 1; r < 5; ++r) c[r] = rsel(app, c[r] - (hc[k] * q[k][r]) * tmp, c[r]);
    }
    // back substitution on the leading nzp x nzp triangle (column oriented)
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        const M act = TR::splat(double(i)) < nzp;
        c[i] = rsel(act, c[i] / q[i][i], c[i]);
#pragma unroll
        for (int r = 0; r < i; ++r) c[r] = rsel(act, c[r] - c[i] * q[i][r], c[r]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        x[j] = zero;
#pragma unroll
        for (int i = 0; i < 3; ++i) x[j] = rsel(pid[i] == TR::splat(double(j)), rsel(TR::splat(double(i)) < nzp, c[i], zero), x[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// point-to-plane residual on 5 neighbours (loam_point_to_plane_ivox.h:275-321), branch-free.
// A[c][r]: coordinate c of neighbour r (row 0 = the nearest), ps: source point (body frame), pt: transformed point
// (already rounded to float and widened again), T: pose, column-major 4x4 (uniform).  Returns the validity mask.
// ---------------------------------------------------------------------------------------------------------------
template <class R>
FLS_HD typename RealTraits<R>::Mask plane_residual_bf(const R (&A)[3][5], const R (&ps)[3], const R (&pt)[3], const double* __restrict__ T,
                                                      const double thres, R (&J)[6], R& d_abs) {
    using TR = RealTraits<R>;
    using M = typename TR::Mask;
    R x[3];
    plane_fit_5x3_bf<R>(A, x);
    const R nrm = rsqrt_((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    M ok = TR::mtrue();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const R r = ((A[0][j] * x[0] + A[1][j] * x[1]) + A[2][j] * x[2]) + TR::splat(1.0);
        ok = ok && !(rabs_(r) / nrm > TR::splat(thres));
    }
    const R n0 = x[0] / nrm, n1 = x[1] / nrm, n2 = x[2] / nrm;
    const R d = ((pt[0] - A[0][0]) * n0 + (pt[1] - A[1][0]) * n1) + (pt[2] - A[2][0]) * n2;
    const R range = rsqrt_((ps[0] * ps[0] + ps[1] * ps[1]) + ps[2] * ps[2]);
    ok = ok && !(range < TR::splat(81.0) * d * d);
    const R s = rsel(d > TR::splat(0.0), TR::splat(1.0), TR::splat(-1.0));
    const R v0 = (TR::splat(T[0]) * ps[0] + TR::splat(T[4]) * ps[1]) + TR::splat(T[8]) * ps[2];
    const R v1 = (TR::splat(T[1]) * ps[0] + TR::splat(T[5]) * ps[1]) + TR::splat(T[9]) * ps[2];
    const R v2 = (TR::splat(T[2]) * ps[0] + TR::splat(T[6]) * ps[1]) + TR::splat(T[10]) * ps[2];
    const R zero = TR::splat(0.0);
    J[0] = ((zero * n0 + (-v2) * n1) + v1 * n2) * s;
    J[1] = ((v2 * n0 + zero * n1) + (-v0) * n2) * s;
    J[2] = (((-v1) * n0 + v0 * n1) + zero * n2) * s;
    J[3] = n0 * s;
    J[4] = n1 * s;
    J[5] = n2 * s;
    d_abs = rabs_(d);
    return ok;
}

}  // namespace fls
