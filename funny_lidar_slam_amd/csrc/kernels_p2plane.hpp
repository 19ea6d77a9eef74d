// kernels_p2plane.hpp -- the point-to-plane correspondence + residual kernel against the
// iVox hash-voxel map, and the LOAM-family Gauss-Newton solve kernel.   gfx950 only.
//
// The iVox correspondence + fit kernels (kernels_ivox_coop.hpp) replace, per Gauss-Newton iteration
// (SURVEY.md 8a rows a4, a5, a9-a11):
//   LoamPointToPlaneIVOX::PlanerMatch           loam_point_to_plane_ivox.h:256-324
//     pcl::transformPoint(double -> float)       :265-266
//     IVoxMap::GetClosestPoint (19-voxel kNN-5)  src/ivox_map/ivox_map.cpp:6-37
//       IVoxMap::Pos2Grid                        src/ivox_map/ivox_map.cpp:145-147
//       VoxelGridNode::KNNPointByCondition       src/ivox_map/voxel_grid_node.cpp:23-42
//       DistanceSquared (float)                  include/common/pointcloud_utility.h:14-17
//     5x3 plane fit + gates + Jacobian           :275-321
//   and the per-wave part of SumCoefficient      :326-340
// gn_solve_loam_kernel finishes SumCoefficient (fixed-order reduction of the wave partials) and runs
//   dx = H.fullPivHouseholderQr().solve(g); R <- Exp(dx[0:3]) R; t += dx[3:6]; stop rule   :167-195
// on the device: the host never synchronises inside a Match, it polls the result mailbox.
//
// This header holds what every LOAM-family kind shares: the point-to-plane residual, the wave reduction of
// the rank-1 normal-equation terms and the Gauss-Newton tail.  The 29 sums of a wave are entries of sum_p v_p v_p^T with
// v_p = (J0..J5, r, 1): they run as eight v_mfma_f64_16x16x4_f64 issues over a per-wave LDS tile (reduce_rank1_mfma_and_store, fixed
// issue order -> bit-reproducible; a measured departure from north_star's "no MFMA" wording, DESIGN.md section 4: fit launch
// 17.4 -> 15.9 us); -DFLS_FIT_MFMA=0 builds the DPP row-shift tree they replaced.  Nothing else in this path is a dense contraction.
#pragma once
#include "host_math.hpp"
#include "device_common.hpp"
#include "linalg_dev.hpp"
#include "wave_solve.hpp"

namespace fls {

// Point-to-plane residual on 5 neighbours (Appendix C.1).  nn[0] must be the nearest neighbour.
// Returns false wherever the reference lambda returns early.
__device__ __forceinline__ bool plane_residual_dev(const float4 (&nn)[5], const float spx, const float spy, const float spz,
                                                   const float ptx, const float pty, const float ptz, const double* __restrict__ T,
                                                   const double thres, double (&J)[6], double& d_abs) {
    double A[3][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { A[0][j] = (double)nn[j].x; A[1][j] = (double)nn[j].y; A[2][j] = (double)nn[j].z; }
    double x[3];
    plane_fit_5x3(A, x);
    const double nrm = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double r = ((A[0][j] * x[0] + A[1][j] * x[1]) + A[2][j] * x[2]) + 1.0;
        if (fabs(r) / nrm > thres) ok = false;
    }
    if (!ok) return false;
    const double n0 = x[0] / nrm, n1 = x[1] / nrm, n2 = x[2] / nrm;
    const double ps0 = (double)spx, ps1 = (double)spy, ps2 = (double)spz;
    const double d = (((double)ptx - A[0][0]) * n0 + ((double)pty - A[1][0]) * n1) + ((double)ptz - A[2][0]) * n2;
    const double range = sqrt((ps0 * ps0 + ps1 * ps1) + ps2 * ps2);
    if (range < 81 * d * d) return false;
    const double s = d > 0 ? 1.0 : -1.0;
    const double v0 = (T[0] * ps0 + T[4] * ps1) + T[8] * ps2;
    const double v1 = (T[1] * ps0 + T[5] * ps1) + T[9] * ps2;
    const double v2 = (T[2] * ps0 + T[6] * ps1) + T[10] * ps2;
    J[0] = ((0.0 * n0 + (-v2) * n1) + v1 * n2) * s;
    J[1] = ((v2 * n0 + 0.0 * n1) + (-v0) * n2) * s;
    J[2] = (((-v1) * n0 + v0 * n1) + 0.0 * n2) * s;
    J[3] = n0 * s;
    J[4] = n1 * s;
    J[5] = n2 * s;
    d_abs = fabs(d);
    return true;
}

// wave-reduce the rank-1 contribution (J J^T, -J r, r, 1) of every lane and store one partial row.
// DPP row-shift tree (wave_solve.hpp): the total of each of the 29 sums lands in lane 63.
__device__ __forceinline__ void reduce_rank1_and_store(const bool contrib, const double (&J)[6], const double r,
                                                       double* __restrict__ partial_row) {
    const int lane = threadIdx.x & 63;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            const double v = wave_sum_dpp(contrib ? J[a] * J[b] : 0.0);
            if (lane == 63) partial_row[k] = v;
            ++k;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double v = wave_sum_dpp(contrib ? (-J[a]) * r : 0.0);
        if (lane == 63) partial_row[21 + a] = v;
    }
    const double sr = wave_sum_dpp(contrib ? r : 0.0);
    const double sc = wave_sum_dpp(contrib ? 1.0 : 0.0);
    if (lane == 63) { partial_row[27] = sr; partial_row[28] = sc; }
}

// The same 29 sums through the FP64 matrix cores (round 3).  With v_p = (J0 .. J5, r, 1) per contributing point (zeros otherwise) the
// wave's contribution is the 8 x 8 matrix sum_p v_p v_p^T: H = its leading 6 x 6 block, g_a = -(a, 6), sum r = (6, 7), count = (7, 7).
// v_mfma_f64_16x16x4_f64 multiplies a 16 x 4 by a 4 x 16 operand per issue: columns 0..7 carry four points, columns 8..15 four more,
// so the two diagonal 8 x 8 blocks of the 16 x 16 accumulator are two partial sums (the off-diagonal blocks are cross terms nobody
// reads) and eight issues cover the wave's 64 points.  The points reach the operand layout (lane l supplies component l & 7 of point
// 8 s + 4 ((l >> 3) & 1) + (l >> 4) in step s) through one 4 KB LDS tile per wave: four 16-byte writes and eight 8-byte reads per lane,
// both contiguous.  Against the DPP tree (29 sums x 6 steps x {2 moves + 1 add} = 522 VALU instructions per wave, a fifth of the fit
// phase) this is ~40 instructions plus eight MFMA issues that overlap the other wave's VALU work.  Deterministic (fixed issue order);
// the products are fused into the accumulation (no intermediate rounding), so H and g differ from the DPP tree's in the last bits
// like any other summation order -- integers (the count) are exact.
#ifndef FLS_FIT_MFMA
#define FLS_FIT_MFMA 1  // 0: the DPP tree (A/B builds)
#endif
typedef double fls_double4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void reduce_rank1_mfma_and_store(const bool contrib, const double (&J)[6], const double r, double* __restrict__ partial_row,
                                                            double* __restrict__ tile /* LDS, 512 doubles of this wave, 64-byte aligned */) {
    const int lane = threadIdx.x & 63;
    const double z = 0.0;
    double v[8];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = contrib ? J[a] : z;
    v[6] = contrib ? r : z;
    v[7] = contrib ? 1.0 : z;
    double2* const dst = reinterpret_cast<double2*>(tile + ((lane >> 3) * 64 + (lane & 3) * 16 + ((lane >> 2) & 1) * 8));
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_double2(v[2 * q], v[2 * q + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fls_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const double x = tile[s * 64 + lane];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    // accumulator: element reg of lane l = D[(l >> 4) + 4 reg][l & 15]; top-left block in lanes with (l & 15) < 8 (regs 0, 1), bottom-right
    // block eight lanes further (regs 2, 3)
    const double h0 = acc[0] + __shfl_down(acc[2], 8, 64);  // row (l >> 4)
    const double h1 = acc[1] + __shfl_down(acc[3], 8, 64);  // row (l >> 4) + 4
    const int j = lane & 15, i0 = lane >> 4;
    if (j < 8) {
        // row i0 (0..3)
        if (j < 6 && i0 <= j) partial_row[i0 * 6 - (i0 * (i0 - 1)) / 2 + (j - i0)] = h0;
        if (j == 6) partial_row[21 + i0] = -h0;
        const int i1 = i0 + 4;  // 4..7
        if (i1 < 6) {
            if (j < 6 && i1 <= j) partial_row[i1 * 6 - (i1 * (i1 - 1)) / 2 + (j - i1)] = h1;
            if (j == 6) partial_row[21 + i1] = -h1;
        } else if (i1 == 6) {
            if (j == 7) partial_row[27] = h1;
        } else if (j == 7) {
            partial_row[28] = h1;
        }
    }
}

// The same for IcpOptimized, whose point contributes THREE rank-1 terms (the rows of its 3 x 6 Jacobian: H = J^T J, B = -J^T e): the lane
// that holds a correspondence (lane 0 of an 8-lane search group) writes the vectors (J_r0 .. J_r5, e_r, [r == 0]) of its three rows into
// three of its group's eight tile slots, the other slots are zero -- 24 vectors per wave, the same eight issues.  Row [27] (sum of |e|, not a
// product of the vector's entries) stays a DPP sum; the caller stores it after this call.
__device__ __forceinline__ void reduce_rank1x3_mfma_and_store(const bool contrib, const double (&J)[18] /* 3 x 6 column-major */, const double (&e)[3],
                                                              double* __restrict__ partial_row, double* __restrict__ tile /* LDS, 512 doubles of this wave */) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    auto slot = [&](const int p) { return reinterpret_cast<double2*>(tile + ((p >> 3) * 64 + (p & 3) * 16 + ((p >> 2) & 1) * 8)); };
    const double z = 0.0;
    if (sub == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double2* const dst = slot(lane + r);
            dst[0] = make_double2(contrib ? J[r + 0] : z, contrib ? J[r + 3] : z);
            dst[1] = make_double2(contrib ? J[r + 6] : z, contrib ? J[r + 9] : z);
            dst[2] = make_double2(contrib ? J[r + 12] : z, contrib ? J[r + 15] : z);
            dst[3] = make_double2(contrib ? e[r] : z, (contrib && r == 0) ? 1.0 : z);
        }
    } else if (sub >= 3) {
        double2* const dst = slot(lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_double2(z, z);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fls_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const double x = tile[s * 64 + lane];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    const double h0 = acc[0] + __shfl_down(acc[2], 8, 64);
    const double h1 = acc[1] + __shfl_down(acc[3], 8, 64);
    const int j = lane & 15, i0 = lane >> 4;
    if (j < 8) {
        if (j < 6 && i0 <= j) partial_row[i0 * 6 - (i0 * (i0 - 1)) / 2 + (j - i0)] = h0;
        if (j == 6) partial_row[21 + i0] = -h0;
        const int i1 = i0 + 4;
        if (i1 < 6) {
            if (j < 6 && i1 <= j) partial_row[i1 * 6 - (i1 * (i1 - 1)) / 2 + (j - i1)] = h1;
            if (j == 6) partial_row[21 + i1] = -h1;
        } else if (i1 == 7 && j == 7) {
            partial_row[28] = h1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fixed-order reduction of partial rows: rows [0,nrows) of `partials`, 32 columns, by a workgroup of
// NT threads = NT/32 row-groups x 32 columns; every thread keeps 16 independent loads in flight per trip
// (the first version waited for one memory round trip per row: 41k cycles for 1800 rows).
// Result in tot[32] (LDS), valid after the call.  Summation order is fixed => reproducible.
// ---------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 1024;
// SC1 = true: the rows were published write-through by OTHER workgroups of the SAME launch (the fused fit + solve
// kernel): read them with relaxed agent-scope loads (sc1: served from memory, never from a stale L1 / L2 line).
// U = loads a thread keeps in flight per trip: one trip covers U * NT / 32 rows.  16 until round 6 -- IcpOptimized's 205 rows on a 256-thread
// workgroup and IncrementalNDT's 457 rows on 512 threads took TWO dependent trips of cache-bypassing loads (2.0 us of the tail by the shader-clock
// stamps, tools/gpu_icp_stamps.py); 32 makes it one.  The order of the additions is the same for every U.
// Row tests (end of round 6): a guarded load is ten instructions (row index, compare, zero, exec mask, branch, 64-bit address, load, exec restore), and the
// waves of a 256-thread tail workgroup are alone on their SIMDs, where an instruction costs 6-8 cycles.  The first (nrows - r0) / NG
// rounds of a trip are in range for EVERY row group -- a uniform number -- so those loads need no per-lane test, and a 32-bit byte offset from the uniform base
// needs no 64-bit address per load: two instructions per load instead of eleven.  Written as three straight-line variants chosen by ONE scalar branch (all U
// rounds unguarded / all but the last four / none: a per-load choice was folded back into a guarded load by the compiler).  Same loads, same additions, same
// order: bit-identical sums (tools/gpu_ab_libs.py: equal pose bits on every kind).  Measured (profiles/r06_aw_reduce_uniform_rounds_ab.log): IcpOptimized
// 229.8 -> 225.9 us per Match (rows read + reduced 3,932 -> 3,480 ticks), LoamFull 236.8 -> 235.6; the 512-thread kernels (two waves per SIMD in the tail
// workgroup: iVox, IncrementalNDT) measured 0.4-0.7 us per Match SLOWER with it and keep the guarded form, as do the 1,024-thread solve launches.
#ifndef FLS_REDUCE_UNIFORM_ROUNDS
#define FLS_REDUCE_UNIFORM_ROUNDS 1  // 0: every load guarded per lane everywhere (A/B builds)
#endif
template <int NT, bool SC1, int U, int F /* rounds without a row test */>
__device__ __forceinline__ void reduce_partials_trip(const double* __restrict__ partials, const int r, const int nrows, const int col, double& acc) {
    constexpr int NG = NT / 32;
    // a 32-bit BYTE offset from the uniform base (rows x 256 bytes: far below 2^32): the load takes it as it is (scalar base + 32-bit lane offset)
    auto load_row = [&](const int rr) -> double {
        const unsigned boff = ((unsigned)rr * (unsigned)kPartialStride + (unsigned)col) * 8u;
        const char* const a = reinterpret_cast<const char*>(partials) + boff;
        if (SC1) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        return *reinterpret_cast<const double*>(a);
    };
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int rr = r + u * NG;
        if (u < F) v[u] = load_row(rr);
        else v[u] = rr < nrows ? load_row(rr) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
}
template <int NT, bool SC1 = false, int U = 16>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, const int nrows, double* tot /*LDS 32*/,
                                                double (*red)[33] /*LDS (NT/32) x 33*/) {
    constexpr int NG = NT / 32;
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double acc = 0.0;
    if constexpr (FLS_REDUCE_UNIFORM_ROUNDS && NT <= 256) {
        for (int r = grp; r < nrows; r += U * NG) {
            const int rounds = (nrows - (r - grp)) / NG;  // uniform: rounds of this trip whose row is in range for every row group
            if (rounds >= U) reduce_partials_trip<NT, SC1, U, U>(partials, r, nrows, col, acc);
            else if (rounds >= U - 4) reduce_partials_trip<NT, SC1, U, U - 4>(partials, r, nrows, col, acc);
            else reduce_partials_trip<NT, SC1, U, 0>(partials, r, nrows, col, acc);
        }
    } else {
        for (int r = grp; r < nrows; r += U * NG) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * NG;
                if (SC1) {
                    v[u] = rr < nrows ? __longlong_as_double((long long)__hip_atomic_load(
                                            (const unsigned long long*)partials + (size_t)rr * kPartialStride + col, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT))
                                      : 0.0;
                } else {
                    v[u] = rr < nrows ? partials[(size_t)rr * kPartialStride + col] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    red[grp][col] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int g = 0; g < NG; ++g) t += red[g][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// Sharded fan-in of a launch's workgroups ("am I the last one to arrive?"): 225-450 arrivals on ONE device-scope counter serialise at
// ~12 ns each and the last arriver waits microseconds behind the others; eight counters (shard = blockIdx & 7, the XCD the workgroup
// normally runs on -- a speed matter only, every atomic is agent scope) take their arrivals in parallel, the last arriver of a shard
// moves on to the top counter, the last of those is THE last.  Counters sit 128 bytes apart and are reset by their last arriver, so
// they are zero at every launch.  Call with one thread after the workgroup's row is published and drained; kTicketWords words.
constexpr int kTicketWords = 32 * 10;  // [0] single counter (shards <= 1) | [32 (1 + s)] shard s | [32 * 9] top counter
__device__ __forceinline__ unsigned fanin_last_arriver(unsigned* __restrict__ ticket, const int shards) {
    if (shards <= 1) {
        const unsigned last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return last;
    }
    const unsigned sh = blockIdx.x & 7u, nsh = (gridDim.x - sh + 7u) >> 3, ntop = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned* const cs = ticket + 32u * (1u + sh);
    unsigned* const ct = ticket + 32u * 9u;
    if (__hip_atomic_fetch_add(cs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nsh - 1u) return 0u;
    __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ntop - 1u) return 0u;
    __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 1u;
}
// publish one partial row write-through (sc1) + drain + ticket: returns (uniformly over the workgroup) whether this workgroup is the last
__device__ __forceinline__ bool publish_row_and_arrive(const double v, const bool has_value, double* __restrict__ partials, unsigned* __restrict__ ticket,
                                                       const int shards, unsigned& s_ticket) {
    if (has_value)
        __hip_atomic_store((unsigned long long*)partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x, (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = fanin_last_arriver(ticket, shards);
    __syncthreads();
    return s_ticket != 0u;
}

#ifdef FLS_TIMING
#define FLS_STAMP(k) do { if (threadIdx.x == 0) st->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FLS_STAMP(k) do { } while (0)
#endif

// Fast path of the Gauss-Newton tails.  The 6x6 normal equations H = sum J J^T are symmetric positive definite whenever the
// scene constrains all six degrees of freedom; one lane then solves H x = g by an unpivoted LDL^T in registers (static
// indices only, ~0.8 us) instead of the wave-cooperative restatement of Eigen's FullPivHouseholderQR (6.2 us) / partial-pivot LU
// inverse.  Same linear system, so the same x up to rounding (observed 1e-13 relative; the pose the Match returns agrees
// with the oracle's to <= 1e-12 instead of <= 1e-14, every per-iteration n_valid / flag / id comparison of the test-suite is
// unchanged).  The Eigen-arithmetic solvers remain the fallback whenever a pivot is not safely positive (rank-deficient or
// badly conditioned systems: there Eigen's rank-revealing behaviour IS the semantics) and can be forced for every system
// with FLS_TAIL_EXACT=1 (launch word bit 23).  Returns false when the caller must run the exact solver.
__device__ __forceinline__ bool ldlt_solve6_lane(const double* __restrict__ H /* 6x6 column-major, LDS */, const double* __restrict__ g, double* __restrict__ x) {
    return hm::ldlt_solve6(H, g, x);  // host_math.hpp (__host__ __device__: tests/host/host_logic_test.cpp checks it on the CPU)
}
// The fast path as the tails call it (whole wave 0, uniform result; launch word bit 23 = FLS_TAIL_EXACT: no fast path).  Default since the end of
// round 6: the factorisation with the matrix rows in lanes 0..5 (wave_solve.hpp::ldlt_solve6_wave); -DFLS_TAIL_LDLT_WAVE=0 builds the one-lane
// form it replaced (A/B: tools/gpu_ab_libs.py).
#ifndef FLS_TAIL_LDLT_WAVE
#define FLS_TAIL_LDLT_WAVE 1
#endif
__device__ __forceinline__ int ldlt_fast_path(const double* __restrict__ H, const double* __restrict__ g, double* __restrict__ x, const unsigned launch_word) {
    if ((launch_word >> 23) & 1u) return 0;
#if FLS_TAIL_LDLT_WAVE
    return ldlt_solve6_wave(H, g, x) ? 1 : 0;
#else
    int fast = 0;
    if ((threadIdx.x & 63) == 0) fast = ldlt_solve6_lane(H, g, x) ? 1 : 0;
    return __shfl(fast, 0, 64);
#endif
}

// shared memory of the LOAM-family Gauss-Newton tail
struct LoamTailSmem {
    double red[32][33];
    double tot_a[32], tot_b[32];
    double Hs[36], gs[6], xs[6], hc[6];
    int tr[6], ctr[6];
};

// LOAM family tail: loam_point_to_plane_ivox.h:160-196 (same in loam_full_kdtree.h:121-176,
// loam_point_to_plane_kdtree.h:99-136).  partials_a (corner rows, may be empty) are summed before
// partials_b (planar rows), as SumCoefficient does (loam_full_kdtree.h:347-372).  Executed by a whole
// workgroup of NT threads: all reduce the rows, wave 0 then runs the cooperative full-pivot Householder QR
// (wave_solve.hpp), lane 0 applies the left-multiplicative update and the stop rule.
// Tl / last_rot / last_pos / it: state words loaded by the caller BEFORE any waiting (latency overlap).
template <int NT, bool SC1 = false>
__device__ __forceinline__ void loam_tail(GnState* __restrict__ st, LoamTailSmem& sm, const double* __restrict__ partials_a,
                                          const int nrows_a, const double* __restrict__ partials_b, const int nrows_b,
                                          const double rot_thr, const double pos_thr, double (&Tl)[16], const double last_rot,
                                          const double last_pos, const int it, Mailbox* __restrict__ mb = nullptr,
                                          const unsigned match_id = 0u) {
    constexpr int RU = NT <= 256 ? 32 : 16;  // (LoamFull's 225 planar rows on a 256-thread workgroup in one trip)
    if (nrows_a > 0) reduce_partials<NT, SC1, RU>(partials_a, nrows_a, sm.tot_a, sm.red);
    else { if (threadIdx.x < 32) sm.tot_a[threadIdx.x] = 0.0; __syncthreads(); }
    FLS_STAMP(2);
    reduce_partials<NT, SC1, RU>(partials_b, nrows_b, sm.tot_b, sm.red);
    if (threadIdx.x >= 64) return;  // wave 0 only from here on
    FLS_STAMP(3);
    const int lane = threadIdx.x;
    if (lane < 36) {
        const int i = lane % 6, j = lane / 6;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const int k = a * 6 - (a * (a - 1)) / 2 + (b - a);  // index of (a,b), a <= b, in the row-wise upper triangle
        const double v = sm.tot_a[k] + sm.tot_b[k];
        sm.Hs[lane] = v;
        st->H[lane] = v;
    }
    if (lane < 6) { const double v = sm.tot_a[21 + lane] + sm.tot_b[21 + lane]; sm.gs[lane] = v; st->g[lane] = v; }
    __builtin_amdgcn_wave_barrier();
    const int fast = ldlt_fast_path(sm.Hs, sm.gs, sm.xs, match_id);
    if (!fast)
#ifdef FLS_TIMING
    fullpiv_qr_solve6_wave(sm.Hs, sm.gs, sm.xs, sm.hc, sm.tr, sm.ctr, st->dbg);
#else
    fullpiv_qr_solve6_wave(sm.Hs, sm.gs, sm.xs, sm.hc, sm.tr, sm.ctr);
#endif
    FLS_STAMP(4);
    if (lane == 0) {
        double dx[6];
        for (int q = 0; q < 6; ++q) dx[q] = sm.xs[q];
        double Rd[9], R[9], Rn[9];
        so3_exp_dev(dx, Rd);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) R[i + j * 3] = Tl[i + j * 4];
        mat3_mul_dev(Rd, R, Rn);  // left-multiplicative
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) Tl[i + j * 4] = Rn[i + j * 3];
        Tl[12] += dx[3];
        Tl[13] += dx[4];
        Tl[14] += dx[5];
        const double rn = norm3d(dx), pn = norm3d(dx + 3);
        const double drot = fabs(rn - last_rot), dpos = fabs(pn - last_pos);
        const int nvb = (int)sm.tot_b[28], nva = (int)sm.tot_a[28];
        const double srb = sm.tot_b[27], sra = sm.tot_a[27];
        for (int q = 0; q < 16; ++q) st->T[q] = Tl[q];
        st->last_rot = rn;
        st->last_pos = pn;
        for (int q = 0; q < 6; ++q) st->last_dx[q] = dx[q];
        st->n_valid = nvb;
        st->n_valid2 = nva;
        st->sum_res = srb;
        st->sum_res2 = sra;
        if (it < kMaxIter) {
            for (int q = 0; q < 16; ++q) st->log_T[it][q] = Tl[q];
            st->log_nv[it] = nvb;
            st->log_res[it] = srb;
        }
        st->iter = it + 1;
        const int stop = ((rn < rot_thr && pn < pos_thr) || (drot < 1.0e-4 && dpos < 1.0e-4)) ? 1 : 0;
        st->done = stop;
        if (mb) {
            mailbox_publish(mb, Tl, dx, srb, sra, it + 1, stop, 0, nvb, nva, match_id);  // launch word: max_iterations << 24 | match id
        }
        FLS_STAMP(5);
    }
}

// test hook (fls_debug_fullpiv_qr6): one wave per 6x6 system, the Gauss-Newton tail's solver on caller-supplied matrices
__global__ void __launch_bounds__(64)
debug_fullpiv_qr6_kernel(const double* __restrict__ H, const double* __restrict__ g, const int n, double* __restrict__ x) {
    const int s = blockIdx.x;
    if (s >= n) return;
    __shared__ double Hs[36], gs[6], xs[6], hc[6];
    __shared__ int tr[6], ctr[6];
    if (threadIdx.x < 36) Hs[threadIdx.x] = H[(size_t)s * 36 + threadIdx.x];
    if (threadIdx.x < 6) gs[threadIdx.x] = g[(size_t)s * 6 + threadIdx.x];
    __builtin_amdgcn_wave_barrier();
    fullpiv_qr_solve6_wave(Hs, gs, xs, hc, tr, ctr);
    if (threadIdx.x < 6) x[(size_t)s * 6 + threadIdx.x] = xs[threadIdx.x];
}

// test hook (fls_debug_ldlt6): one wave per 6x6 system, the fast path exactly as the tails call it; ok[s] = 1 where the fast path accepted the system
__global__ void __launch_bounds__(64)
debug_ldlt6_kernel(const double* __restrict__ H, const double* __restrict__ g, const int n, double* __restrict__ x, int* __restrict__ ok) {
    const int s = blockIdx.x;
    if (s >= n) return;
    __shared__ double Hs[36], gs[6], xs[6];
    if (threadIdx.x < 36) Hs[threadIdx.x] = H[(size_t)s * 36 + threadIdx.x];
    if (threadIdx.x < 6) { gs[threadIdx.x] = g[(size_t)s * 6 + threadIdx.x]; xs[threadIdx.x] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    const int fast = ldlt_fast_path(Hs, gs, xs, 0u);
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x < 6) x[(size_t)s * 6 + threadIdx.x] = fast ? xs[threadIdx.x] : 0.0;
    if (threadIdx.x == 0) ok[s] = fast;
}

__global__ void __launch_bounds__(kSolveThreads)
gn_solve_loam_kernel(GnState* __restrict__ st, const int first, const Pose16 T0, const double* __restrict__ partials_a,
                     const int nrows_a, const double* __restrict__ partials_b, const int nrows_b, const double rot_thr,
                     const double pos_thr, Mailbox* __restrict__ mb, const unsigned match_id) {
    // every state word the tail needs is loaded up front (latency overlaps the partial reduction); the pose waits in LDS, not in 32 registers of
    // every lane: at 1,024 threads the kernel has 128 VGPRs and spilled 17 words to scratch with the pose live across the solve
    const int done = first ? 0 : st->done;
    __shared__ double Tl[16];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) Tl[q] = first ? T0.m[q] : st->T[q];
    }
    const double last_rot = first ? 0.0 : st->last_rot, last_pos = first ? 0.0 : st->last_pos;
    const int it = first ? 0 : st->iter;
    if (done) return;
    __shared__ LoamTailSmem sm;
    loam_tail<kSolveThreads>(st, sm, partials_a, nrows_a, partials_b, nrows_b, rot_thr, pos_thr, Tl, last_rot, last_pos, it, mb, match_id);
}

}  // namespace fls
