// Round-2 experiment (NOT compiled into the product): column-per-lane layout of the 6x6 full-pivot Householder QR.
// Bit-exact against the oracle on tests/test_gpu_solver.py's 3,000 systems, but 1 us per Gauss-Newton iteration SLOWER than the
// element-per-lane version in wave_solve.hpp (Match 128.8-129.2 us vs 125.7-126.0 us): ~200 v_readlane hand-offs cost more than
// the LDS round trips they replace.  See tools/experiments/README.md.
// ---------------------------------------------------------------------------------------------
// fullpiv_qr_solve6_cols -- the same Eigen FullPivHouseholderQR<6x6>::solve semantics (pivot order, rank rule, zero fill; the
// arithmetic of oracle/flo_linalg.h fullpiv_qr_solve<6>, operation for operation), laid out for a short dependency chain:
//   * lane c < 6 keeps COLUMN c of H in six registers, lane 6 keeps the right-hand side g (it is row-swapped and reflected
//     along with the trailing columns, so c = Q^T g is finished when the factorisation is -- Eigen applies the same swaps and
//     reflectors to the rhs inside solve(); the ones of steps >= rank only touch rows that solve() never reads);
//   * columns never move between lanes: the column permutation is a uniform position -> lane table; row swaps are register
//     swaps inside every lane; a step exchanges one column through v_readlane and nothing through LDS;
//   * the reflector's divisions (up to five essential-part entries and tau) are ONE division instruction sequence: lane r
//     divides entry r, lane 7 computes tau, the results go back by v_readlane;
//   * the triangular solve runs on uniform values (v_readlane of R's columns).
// v1 (above) cost 10.3k cycles for the factorisation + 4.5k for the solve phase per Gauss-Newton iteration (tools/experiments).
// x[6] returns the solution in every lane.  Must be called by all 64 lanes of one wave.  H is symmetric on entry: lane c's column
// is also row c, but nothing here relies on that.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int uniform_i(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double readlane_dyn_f64(const double v, const int lane /* uniform */) {
    const int l = uniform_i(lane);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// a[i] with a run-time uniform index, as a select chain (no indexed private arrays -> no scratch)
__device__ __forceinline__ int pick6(const int (&a)[6], const int i) {
    int r = a[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) r = (i == q) ? a[q] : r;
    return r;
}

__device__ __forceinline__ void fullpiv_qr_solve6_cols(const double* __restrict__ Hs /* 36, column-major (LDS or global) */,
                                                       const double* __restrict__ gs /* 6 */, double (&xout)[6]) {
    const int lane = threadIdx.x & 63;
    double x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) x[r] = lane < 6 ? Hs[r + 6 * lane] : (lane == 6 ? gs[r] : 0.0);
    int pos2lane[6] = {0, 1, 2, 3, 4, 5};  // uniform: which lane holds the column at position p
    int mypos = lane < 6 ? lane : (lane == 6 ? 6 : 7);  // position of this lane's column (6: the rhs, 7: idle lanes)
    double diag[6] = {0, 0, 0, 0, 0, 0};    // uniform: R(k, k)
    const double precision = FLS_DBL_EPS * 6.0;
    double biggest = 0.0, maxpivot = 0.0;
    int nzp = 6;
    bool stopped = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!stopped) {
            // 1. pivot: biggest |entry| of the bottom-right corner, first maximum in column-major POSITION order
            double cm = -1.0;
            int cr = k;
#pragma unroll
            for (int r = k; r < 6; ++r) {
                const double a = fabs(x[r]);
                const bool gt = a > cm;
                cm = gt ? a : cm;
                cr = gt ? r : cr;
            }
            double bc = -1.0;
            int cbpos = k, rb = k;
#pragma unroll
            for (int pp = k; pp < 6; ++pp) {
                const int ln = pos2lane[pp];
                const double m = readlane_dyn_f64(cm, ln);
                const int rr = __builtin_amdgcn_readlane(cr, uniform_i(ln));
                const bool gt = m > bc;
                bc = gt ? m : bc;
                cbpos = gt ? pp : cbpos;
                rb = gt ? rr : rb;
            }
            if (k == 0) biggest = bc;
            if (fabs(bc) <= fabs(biggest) * precision) {  // isMuchSmallerThan: the rest of the corner is negligible
                nzp = k;
                stopped = true;
            } else {
                // 2. column swap = the position table; row swap k <-> rb inside every column at a position >= k, and in the rhs
                const int lane_k = pos2lane[k], lane_cb = pick6(pos2lane, cbpos);
#pragma unroll
                for (int q = 0; q < 6; ++q) pos2lane[q] = (q == k) ? lane_cb : ((q == cbpos) ? lane_k : pos2lane[q]);
                if (lane == lane_cb) mypos = k;
                else if (lane == lane_k) mypos = cbpos;
                const bool live = mypos >= k && mypos <= 6;
#pragma unroll
                for (int r = k + 1; r < 6; ++r) {
                    const bool sw = live && rb == r;
                    const double a = x[k], b = x[r];
                    x[k] = sw ? b : a;
                    x[r] = sw ? a : b;
                }
                // 3. Householder reflector of the pivot column (rows k..5): every lane evaluates its own column, the pivot
                //    lane's numbers are the ones broadcast
                double tail = 0.0;
#pragma unroll
                for (int r = k + 1; r < 6; ++r) tail += x[r] * x[r];
                const double c0 = x[k];
                const bool trivial = (k == 5) || tail <= FLS_DBL_MIN;
                double beta = sqrt(c0 * c0 + tail);
                if (c0 >= 0.0) beta = -beta;
                const int P = lane_cb;
                const double beta_p = readlane_dyn_f64(beta, P), c0_p = readlane_dyn_f64(c0, P);
                const bool trivial_p = __builtin_amdgcn_readlane(trivial ? 1 : 0, uniform_i(P)) != 0;
                // one division sequence for all quotients: lane r (k < r < 6) divides entry r by (c0 - beta), lane 7 makes tau
                double num = 0.0, den = 1.0;
#pragma unroll
                for (int r = k + 1; r < 6; ++r) {
                    const double xr = readlane_dyn_f64(x[r], P);
                    if (lane == r) { num = xr; den = c0_p - beta_p; }
                }
                if (lane == 7) { num = beta_p - c0_p; den = beta_p; }
                const double quo = num / den;
                const double tau_u = trivial_p ? 0.0 : readlane_f64(quo, 7);
                const double beta_u = trivial_p ? c0_p : beta_p;
                double ess[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int r = k + 1; r < 6; ++r) ess[r] = trivial_p ? 0.0 : readlane_f64(quo, r);
                diag[k] = beta_u;
                if (fabs(beta_u) > maxpivot) maxpivot = fabs(beta_u);
                if (lane == P) {
                    x[k] = beta_u;
#pragma unroll
                    for (int r = k + 1; r < 6; ++r) x[r] = ess[r];
                } else if (k < 5 && tau_u != 0.0 && mypos > k && mypos <= 6) {
                    // 4. apply to the trailing columns and to the rhs (apply_householder_left, one column each)
                    double tmp = 0.0;
#pragma unroll
                    for (int r = k + 1; r < 6; ++r) tmp += ess[r] * x[r];
                    tmp += x[k];
                    x[k] -= tau_u * tmp;
#pragma unroll
                    for (int r = k + 1; r < 6; ++r) x[r] -= (tau_u * ess[r]) * tmp;
                }
            }
        }
    }
    // rank(): pivots above eps * 6 * |maxpivot|
    const double premult = fabs(maxpivot) * (FLS_DBL_EPS * 6.0);
    int rank = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) rank += (q < nzp && fabs(diag[q]) > premult) ? 1 : 0;
    // c = Q^T g sits in lane 6; back substitution on the leading rank x rank triangle (column oriented), uniform values
    double c[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) c[q] = readlane_f64(x[q], 6);
#pragma unroll
    for (int q = 5; q >= 0; --q) {
        if (q < rank) {
            c[q] /= diag[q];
            const int ln = pos2lane[q];
#pragma unroll
            for (int r = 0; r < q; ++r) c[r] -= c[q] * readlane_dyn_f64(x[r], ln);
        }
    }
    // x[perm[q]] = c[q] for q < rank, 0 elsewhere; perm[q] = the original column at position q = pos2lane[q]
#pragma unroll
    for (int m = 0; m < 6; ++m) xout[m] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < 6; ++m)
            if (q < rank && pos2lane[q] == m) xout[m] = c[q];
}

