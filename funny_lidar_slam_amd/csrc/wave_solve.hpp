// wave_solve.hpp -- wavefront-cooperative pieces of the Gauss-Newton tail (gfx950, wave64):
//
//   wave_sum_dpp          64-lane FP64 sum with DPP row shifts / row broadcasts (no LDS round trips):
//                         row_shr 1,2,4,8 -> row_bcast15 -> row_bcast31; total lands in lane 63.
//                         Fixed tree => bit-reproducible.  Replaces the ds_bpermute shuffle chain that
//                         cost ~20k cycles per wave for the 29 sums of the normal equations.
//   fullpiv_qr_solve6_wave  Eigen FullPivHouseholderQR<6x6>::solve semantics (pivot order, rank rule,
//                         zero fill) executed by ONE wave: lane l < 36 owns element (l % 6, l / 6) of the
//                         LDS-resident matrix; pivot search, row/column swaps and the Householder update of
//                         all trailing columns happen in parallel; the rhs lives in lanes 0..5 and moves
//                         by v_readlane.  Same arithmetic expressions, same summation order as
//                         the scalar algorithm of the CPU path -- only the single-thread LDS latency chain
//                         (~29 us per iteration in the first version) is gone.
//   lu6_solve_wave        PartialPivLU<6x6>: determinant, explicit inverse, inverse * b for ICP / NDT.
#pragma once
#include "linalg_dev.hpp"

namespace fls {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
// after the call lane 63 holds the sum over all 64 lanes (other lanes hold partial prefix sums)
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max(const double v) {  // operands are >= -1: a zero fill never wins wrongly
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(__double2loint(-1.0), lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(__double2hiint(-1.0), hi, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(hi2, lo2);
    return o > v ? o : v;
}
// maximum over the 64 lanes, valid in lane 63 (same tree as wave_sum_dpp)
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = dpp_max<0x111, 0xf>(v);
    v = dpp_max<0x112, 0xf>(v);
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    return v;
}

// counters of the traffic-counting kernel variants: exact integers in doubles, total in every lane
__device__ __forceinline__ double wave_sum_u(const double v) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wave_sum_dpp(v)), 63), __builtin_amdgcn_readlane(__double2loint(wave_sum_dpp(v)), 63)); }

__device__ __forceinline__ double readlane_f64(const double v, const int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// M: LDS, 36 doubles column-major, holds H on entry (destroyed).  g_in: rhs (uniform pointer, 6 doubles).
// Result x[6] is written to xs (LDS, 6 doubles).
// Must be called by all 64 lanes of exactly one wave.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fullpiv_qr_solve6_wave(double* M, const double* g_in, double* xs, double* hcoef /*LDS 6*/,
                                              int* rows_tr /*LDS 6*/, int* cols_tr /*LDS 6*/, long long* dbg = nullptr) {
#define QR_STAMP(q) do { if (dbg && (threadIdx.x & 63) == 0) dbg[q] = (long long)__builtin_readcyclecounter(); } while (0)
    const int lane = threadIdx.x & 63;
    const int l = lane < 36 ? lane : 35;
    const int i = l % 6, j = l / 6;
    const double precision = FLS_DBL_EPS * 6.0;
    double biggest = 0.0, maxpivot = 0.0;
    int nonzero_pivots = 6;
    double cur = lane < 36 ? M[l] : 0.0;  // this lane's element, kept in a register across the steps
    for (int k = 0; k < 6; ++k) {
        // 1. pivot: largest |entry| of the bottom-right corner, first in column-major order on ties
        const double a = (lane < 36 && i >= k && j >= k) ? fabs(cur) : -1.0;
        const double mx = readlane_f64(wave_max_dpp(a), 63);
        if (k == 0) QR_STAMP(6);
        const unsigned long long eq = __ballot(a == mx);
        const int piv = __ffsll((long long)eq) - 1;
        const int rb = piv % 6, cb = piv / 6;
        if (k == 0) biggest = mx;
        if (fabs(mx) <= fabs(biggest) * precision) {  // isMuchSmallerThan: the rest of the corner is negligible
            nonzero_pivots = k;
            if (lane >= k && lane < 6) { rows_tr[lane] = lane; cols_tr[lane] = lane; hcoef[lane] = 0.0; }
            break;
        }
        if (lane == 0) { rows_tr[k] = rb; cols_tr[k] = cb; }
        // 2. the row swap (k <-> rb, columns >= k) and the column swap (k <-> cb, all rows) are not materialised:
        //    the pivoted matrix N(r, c) = old(rowp(r, colp(c)), colp(c)) is read straight from the old one.
        //    One batch of LDS reads: column k (uniform addresses -> broadcast) and this lane's own column j.
        const int jp = (j == k) ? cb : (j == cb ? k : j);
        double ck[6], cj[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int rp = (r == k) ? rb : (r == rb ? k : r);  // row partner (applies to columns >= k)
            ck[r] = M[rp + 6 * cb];                            // column k of N comes from old column cb (>= k)
            cj[r] = M[((jp >= k) ? rp : r) + 6 * jp];
        }
        if (k == 0) QR_STAMP(7);
        double mine = 0.0, cjk = 0.0, c0 = 0.0, tail = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r == i) mine = cj[r];
            if (r == k) { cjk = cj[r]; c0 = ck[r]; }
            if (r > k) tail += ck[r] * ck[r];
        }
        if (k == 0) QR_STAMP(8);
        // 3. Householder reflector of column k (rows k..5), computed redundantly by every lane
        double tau, beta, den = 1.0;
        const bool trivial = (k == 5) || tail <= FLS_DBL_MIN;
        if (trivial) { tau = 0.0; beta = c0; }
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            den = c0 - beta;
            tau = (beta - c0) / beta;
        }
        if (k == 0) QR_STAMP(9);
        // essential part: lane (r, k) divides once, everybody picks the values up with v_readlane (uniform index)
        const double ess_own = (j == k && i > k && !trivial) ? mine / den : 0.0;
        double ess[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) ess[r] = readlane_f64(ess_own, r + 6 * k);  // 0 for r <= k
        // 4. new value of this lane's element
        if (j == k) {
            if (i == k) mine = beta;
            else if (i > k) mine = ess_own;
        } else if (j > k && i >= k && tau != 0.0) {
            double tmp = 0.0, ess_i = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (r > k) tmp += ess[r] * cj[r];
                if (r == i) ess_i = ess[r];
            }
            tmp += cjk;
            if (i == k) mine -= tau * tmp;
            else mine -= (tau * ess_i) * tmp;
        }
        if (k == 0) QR_STAMP(10);
        __builtin_amdgcn_wave_barrier();
        if (lane < 36 && j >= k) { M[l] = mine; cur = mine; }
        if (lane == 0) hcoef[k] = tau;
        __builtin_amdgcn_wave_barrier();
        if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
    }
    QR_STAMP(11);
    // ---- solve phase: uniform work, done redundantly by every lane on registers (no cross-lane traffic, one
    // batch of broadcast LDS reads).  Loops are fully unrolled; run-time pivot indices become predicated swaps.
    __builtin_amdgcn_wave_barrier();
    double A[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) A[q] = M[q];
    double hcf[6];
    int rtr[6], ctr[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { hcf[q] = hcoef[q]; rtr[q] = rows_tr[q]; ctr[q] = cols_tr[q]; }
    // rank(): pivots above eps * 6 * |maxpivot|
    const double premult = fabs(maxpivot) * (FLS_DBL_EPS * 6.0);
    int rank = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) rank += (q < nonzero_pivots && fabs(A[q + 6 * q]) > premult) ? 1 : 0;
    // column permutation: identity, then transposition (k, cols_tr[k]) for k = 0..5
    int perm[6] = {0, 1, 2, 3, 4, 5};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int m = k + 1; m < 6; ++m)  // cols_tr[k] >= k always
            if (ctr[k] == m) { const int t = perm[k]; perm[k] = perm[m]; perm[m] = t; }
    }
    // c = Q^T g
    double c[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) c[q] = g_in[q];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < rank) {
#pragma unroll
            for (int m = k + 1; m < 6; ++m)  // rows_tr[k] >= k always
                if (rtr[k] == m) { const double t = c[k]; c[k] = c[m]; c[m] = t; }
            const double tau = hcf[k];
            if (k == 5) c[5] *= (1.0 - tau);  // rows == 1 case of applyHouseholderOnTheLeft
            else if (tau != 0.0) {
                double tmp = 0.0;
#pragma unroll
                for (int r = k + 1; r < 6; ++r) tmp += A[r + 6 * k] * c[r];
                tmp += c[k];
                c[k] -= tau * tmp;
#pragma unroll
                for (int r = k + 1; r < 6; ++r) c[r] -= (tau * A[r + 6 * k]) * tmp;
            }
        }
    }
    // back substitution on the leading rank x rank triangle (column oriented)
#pragma unroll
    for (int q = 5; q >= 0; --q) {
        if (q < rank) {
            c[q] /= A[q + 6 * q];
#pragma unroll
            for (int r = 0; r < q; ++r) c[r] -= c[q] * A[r + 6 * q];
        }
    }
    // x[perm[q]] = c[q] for q < rank, 0 elsewhere
    double x[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < 6; ++m)
            if (q < rank && perm[q] == m) x[m] = c[q];
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) xs[q] = x[q];
    }
    __builtin_amdgcn_wave_barrier();
    QR_STAMP(12);
#undef QR_STAMP
}

// ---------------------------------------------------------------------------------------------
// PartialPivLU<6x6> by one wave.  M: LDS 36 (H on entry, LU on exit), inv: LDS 36 (out), b: 6 doubles.
// Returns the determinant (uniform); xs[6] (LDS) = inverse * b, summed in column order like the CPU path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double lu6_solve_wave(double* M, double* inv, const double* b, double* xs, int* row_tr /*LDS 6*/) {
    const int lane = threadIdx.x & 63;
    const int l = lane < 36 ? lane : 35;
    const int i = l % 6, j = l / 6;
    int ntr = 0;
    for (int k = 0; k < 6; ++k) {
        // partial pivot in column k, first maximum wins
        int rb = k;
        double bc = fabs(M[k + 6 * k]);
        for (int r = k + 1; r < 6; ++r) { const double v = fabs(M[r + 6 * k]); if (v > bc) { bc = v; rb = r; } }
        if (lane == 0) row_tr[k] = rb;
        if (bc != 0.0 && rb != k) {
            const int ip = (i == k) ? rb : (i == rb ? k : i);
            const double moved = M[ip + 6 * j];
            __builtin_amdgcn_wave_barrier();
            if (lane < 36) M[l] = moved;
            __builtin_amdgcn_wave_barrier();
            ++ntr;
        }
        double mine = M[l];
        const double pivot = M[k + 6 * k];
        if (bc != 0.0 && j == k && i > k) mine = mine / pivot;
        // trailing update uses the scaled column: lu(i,j) -= lu(i,k) * lu(k,j)
        if (j > k && i > k) {
            const double lik = (bc != 0.0) ? M[i + 6 * k] / pivot : M[i + 6 * k];
            mine -= lik * M[k + 6 * j];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 36) M[l] = mine;
        __builtin_amdgcn_wave_barrier();
    }
    double det = M[0];
    for (int q = 1; q < 6; ++q) det *= M[q + 6 * q];
    det = (ntr & 1) ? -det : det;
    // inverse: lanes 0..5 each solve one column of P * I
    if (lane < 6) {
        double c[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) c[r] = (r == lane) ? 1.0 : 0.0;
        // apply the row transpositions to the identity column e_lane: rows swap k <-> row_tr[k] in order
        for (int k = 0; k < 6; ++k) {
            const int rt = row_tr[k];
            if (rt != k) {
                double a = 0.0, bb = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) { if (r == k) a = c[r]; if (r == rt) bb = c[r]; }
#pragma unroll
                for (int r = 0; r < 6; ++r) { if (r == k) c[r] = bb; else if (r == rt) c[r] = a; }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int r = q + 1; r < 6; ++r) c[r] -= c[q] * M[r + 6 * q];
#pragma unroll
        for (int q = 5; q >= 0; --q) {
            c[q] /= M[q + 6 * q];
#pragma unroll
            for (int r = 0; r < q; ++r) c[r] -= c[q] * M[r + 6 * q];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) inv[r + 6 * lane] = c[r];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) {
        double s = 0.0;
        for (int q = 0; q < 6; ++q) s += inv[lane + 6 * q] * b[q];
        xs[lane] = s;
    }
    __builtin_amdgcn_wave_barrier();
    return det;
}

// ---------------------------------------------------------------------------------------------
// ldlt_solve6_wave: the SPD fast path of the Gauss-Newton tails (what hm::ldlt_solve6 does on one lane -- 425 instructions that a lone
// wave issues at ~8 cycles each, 1.45 us of every iteration by the shader-clock stamps) with the ROWS of the working matrix in lanes
// 0..5 (round 6).  Right-looking: at step j the pivot row leaves lane j through v_readlane (SGPRs), every lane scales its own column
// entry and updates its own row -- the 15 trailing-row updates of a step are two instructions instead of thirty -- and the reciprocal of
// the pivot is v_rcp_f64 + two Newton steps (the front half of the IEEE division sequence, <= 1 ulp) instead of a 13-instruction division.
// The pivot rows are uniform values, so U = D^-1 (pivot rows) is known to every lane and the two triangular solves run uniformly without
// further exchange.  Same matrix, same pivots up to rounding (the factorisation is the outer-product form of the same LDL^T); same
// acceptance rule (every pivot > 0, d_min > 1e-9 d_max, finite solution) -- false sends the caller to the restated Eigen solver.
// H: LDS, 6x6 column-major, symmetric, NOT modified.  xs: LDS, written by lane 0.  All 64 lanes of one wave must call.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton_f64(const double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ bool ldlt_solve6_wave(const double* __restrict__ H, const double* __restrict__ g, double* __restrict__ xs) {
    const int lane = threadIdx.x & 63;
    const int row = lane < 6 ? lane : 5;  // (lanes 6.. repeat row 5: their values are never read)
    double a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = H[row + 6 * k];
    double U[6][6], inv[6];
    double dmax = 0.0, dmin = 1.0e300;
    bool positive = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double p[6];
#pragma unroll
        for (int k = j; k < 6; ++k) p[k] = readlane_f64(a[k], j);
        const double d = p[j];
        positive = positive && (d > 0.0);  // false for NaN as well
        dmax = d > dmax ? d : dmax;
        dmin = d < dmin ? d : dmin;
        const double iv = rcp_newton_f64(d);
        inv[j] = iv;
        const double l = lane > j ? a[j] * iv : 0.0;  // L(row, j); rows <= j keep what they hold (row j IS the pivot row)
#pragma unroll
        for (int k = j + 1; k < 6; ++k) {
            U[j][k] = p[k] * iv;  // = L(k, j)
            a[k] -= l * p[k];
        }
    }
    if (!positive || !(dmin > 1.0e-9 * dmax)) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= U[k][i] * y[k];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = y[i] * inv[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= U[i][k] * y[k];
        y[i] = s;
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) finite = finite && (y[i] - y[i] == 0.0);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) xs[i] = y[i];
    }
    __builtin_amdgcn_wave_barrier();
    return finite;  // a non-finite right-hand side goes through the exact solver as well
}

}  // namespace fls
