// kernels_ivox_lds.hpp -- EXPERIMENT (FLS_IVOX_LDS=1, off by default): the LDS-staged variant of ivox_knn_kernel that
// BASELINE.json's north_star sketches and the round-1 verdict asked to try: a workgroup handles 64 ring-adjacent queries, the
// UNION of the voxels they hit is staged once in LDS (each voxel's points copied by the lane that claimed it in an LDS hash
// set), and the candidate loop reads LDS instead of gathering through L1.  Same selection, same keys, same outputs as the
// production kernel (parity-tested through the same tests with the switch on).  Measured A/B: tools/experiments/README.md.
//
// Differences from ivox_knn_kernel<4, false, true, FIRST, true>: after the cell lookups every hit voxel is inserted into
// an LDS hash set keyed by its point-array offset; the winner of an insert reserves staging room and copies the voxel's points
// as {x, y, z, slot bits} (the map slot rides in w: the selection key needs it); a workgroup whose union does not fit
// (kStage points) falls back to the global gather for all its queries.
#pragma once
#include "kernels_ivox_coop.hpp"

namespace fls {

constexpr int kLdsStage = 512;  // staged points per workgroup (8 KB)
constexpr int kLdsHash = 256;   // hash-set entries per workgroup (voxels of 64 neighbouring queries: typically 20-60)

template <bool FIRST>
__global__ void __launch_bounds__(256)
ivox_knn_lds_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                    const GnState* __restrict__ st, const Pose16 T0, const DevGrid grid, const DenseWindow win, const float inv_res,
                    float4* __restrict__ nn_pts /* [n][5] */, unsigned char* __restrict__ nn_cnt, unsigned char* __restrict__ flag, const int chunk) {
    constexpr int G = 4, QPB = 256 / G, R = 5;
    __shared__ __attribute__((aligned(16))) unsigned s_end[QPB][24];
    __shared__ __attribute__((aligned(16))) unsigned s_off[QPB][24];
    __shared__ __attribute__((aligned(16))) float4 stage[kLdsStage];
    __shared__ unsigned hs_key[kLdsHash], hs_off[kLdsHash];
    __shared__ unsigned n_staged, overflow;
    const int nb = (n + QPB - 1) / QPB;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int lb = ((seq / chunk) * 8 + xcd) * chunk + (seq % chunk);
    const int sub = threadIdx.x % G, g = threadIdx.x / G;
    const int q = lb * QPB + g;
    const bool active = lb < nb && q < n;
    const int done = FIRST ? 0 : st->done;
    double T[12];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) T[c * 3 + r] = FIRST ? T0.m[c * 4 + r] : st->T[c * 4 + r];
    const int qq = active ? q : 0;
    const float px = sx[qq], py = sy[qq], pz = sz[qq];
    if (done) return;  // uniform over the launch
    if (FIRST && active && sub == 0) flag[q] = 0;
    hs_key[threadIdx.x & (kLdsHash - 1)] = 0u;
    if (threadIdx.x == 0) { n_staged = 0u; overflow = 0u; }
    const double x = px, y = py, z = pz;
    const float ptx = (float)(((T[0] * x + T[3] * y) + T[6] * z) + T[9]);
    const float pty = (float)(((T[1] * x + T[4] * y) + T[7] * z) + T[10]);
    const float ptz = (float)(((T[2] * x + T[5] * y) + T[8] * z) + T[11]);
    const float fx = roundf(ptx * inv_res), fy = roundf(pty * inv_res), fz = roundf(ptz * inv_res);
    const bool in_range = active && fabsf(fx) < (float)kKeyLimit && fabsf(fy) < (float)kKeyLimit && fabsf(fz) < (float)kKeyLimit;
    const int kx = in_range ? (int)fx : 0, ky = in_range ? (int)fy : 0, kz = in_range ? (int)fz : 0;
    auto cell = [&](const int r, unsigned& beg, unsigned& cnt) {
        const int k = sub + G * r;
        int ox, oy, oz;
        nearby18(k < 19 ? k : 0, ox, oy, oz);
        const int cx = kx + ox - win.ox, cy = ky + oy - win.oy, cz = kz + oz - win.oz;
        const bool ok = in_range && k < 19 && (unsigned)cx < (unsigned)win.nx && (unsigned)cy < (unsigned)win.ny && (unsigned)cz < (unsigned)win.nz;
        const uint2 e = win.cells[ok ? (unsigned)((cz * win.ny + cy) * win.nx + cx) : 0u];
        beg = e.x;
        cnt = ok ? e.y : 0u;
    };
    unsigned b[R], c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cell(r, b[r], c[r]);
    __syncthreads();  // hash set cleared
    // ---- phase A: claim / find every hit voxel in the workgroup's hash set; the claimer reserves staging room
    unsigned hidx[R];
    bool won[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        hidx[r] = 0u;
        won[r] = false;
        if (c[r]) {
            const unsigned key = b[r] + 1u;
            unsigned h = (b[r] * 2654435761u) >> 24;  // 8 bits
            for (int probes = 0;; ++probes) {
                if (probes == kLdsHash) { overflow = 1u; break; }  // more distinct voxels than entries: global gather for this workgroup
                const unsigned old = atomicCAS(&hs_key[h], 0u, key);
                if (old == 0u) {
                    const unsigned off = atomicAdd(&n_staged, c[r]);
                    hs_off[h] = off;
                    if (off + c[r] > (unsigned)kLdsStage) overflow = 1u;
                    won[r] = true;
                    break;
                }
                if (old == key) break;
                h = (h + 1u) & (kLdsHash - 1);
            }
            hidx[r] = h;
        }
    }
    __syncthreads();
    const bool staged = overflow == 0u;  // uniform
    // ---- phase B: the claimers copy their voxels' points {x, y, z, slot}
    if (staged) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (won[r]) {
                const unsigned off = hs_off[hidx[r]];
                for (unsigned k = 0; k < c[r]; ++k) {
                    const float4 p = grid.pts[b[r] + k];
                    stage[off + k] = make_float4(p.x, p.y, p.z, __uint_as_float(b[r] + k));
                }
            }
        }
    }
    __syncthreads();
    // ---- phase C: balanced split table; the offsets are LDS offsets when staged, map slots otherwise
    const double kNone = __hiloint2double((int)kKeyNoneHi, -1);
    double t5[5] = {kNone, kNone, kNone, kNone, kNone};
    auto consider = [&](const float4 p, const unsigned s, const bool ok) {
        const float dx = p.x - ptx, dy = p.y - pty, dz = p.z - ptz;
        const float d2 = dx * dx + (dy * dy + dz * dz);
        top5_insert_dkey(t5, make_dkey(d2, s, ok && d2 < 25.0f));
    };
    unsigned o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = (staged && c[r]) ? hs_off[hidx[r]] : b[r];
    const unsigned tot = ((c[0] + c[1]) + (c[2] + c[3])) + c[4];
    const unsigned nz = (c[0] ? 1u : 0u) + (c[1] ? 1u : 0u) + (c[2] ? 1u : 0u) + (c[3] ? 1u : 0u) + (c[4] ? 1u : 0u);
    const unsigned packed = tot * 32u + nz;
    const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x00, 0xf, 0xf, true);
    const unsigned t1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x55, 0xf, 0xf, true);
    const unsigned t2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xAA, 0xf, 0xf, true);
    const unsigned t3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xFF, 0xf, 0xf, true);
    const unsigned before = (sub > 0 ? t0 : 0u) + (sub > 1 ? t1 : 0u) + (sub > 2 ? t2 : 0u), all = t0 + t1 + t2 + t3;
    const unsigned TOT = all >> 5;
    unsigned pos = before & 31u, run = before >> 5;
    {
        uint2* const rowp = reinterpret_cast<uint2*>(&s_end[g][6 * sub]);
        rowp[0] = make_uint2(~0u, ~0u); rowp[1] = make_uint2(~0u, ~0u); rowp[2] = make_uint2(~0u, ~0u);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (c[r]) { s_off[g][pos] = o[r] - run; run += c[r]; s_end[g][pos] = run; ++pos; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const unsigned Q = (TOT + 3u) >> 2, a = sub * Q, e = a + Q < TOT ? a + Q : TOT;
        unsigned k = 0;
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const uint4 e4 = *reinterpret_cast<const uint4*>(&s_end[g][4 * v]);
            k += (e4.x <= a ? 1u : 0u) + (e4.y <= a ? 1u : 0u) + (e4.z <= a ? 1u : 0u) + (e4.w <= a ? 1u : 0u);
        }
        for (unsigned idx = a; idx < e; idx += 4) {
            const unsigned E0 = s_end[g][k], E1 = s_end[g][k + 1], E2 = s_end[g][k + 2], E3 = s_end[g][k + 3];
            const unsigned O0 = s_off[g][k], O1 = s_off[g][k + 1], O2 = s_off[g][k + 2], O3 = s_off[g][k + 3], O4 = s_off[g][k + 4];
            const unsigned last = e - 1;
            const unsigned i1 = idx + 1, i2 = idx + 2, i3 = idx + 3;
            const unsigned s0 = slot_select<5>(idx, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s1 = slot_select<5>(i1 < last ? i1 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s2 = slot_select<5>(i2 < last ? i2 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s3 = slot_select<5>(i3 < last ? i3 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            if (staged) {  // uniform
                const float4 q0 = stage[s0], q1 = stage[s1], q2 = stage[s2], q3 = stage[s3];
                consider(q0, __float_as_uint(q0.w), true);
                consider(q1, __float_as_uint(q1.w), i1 <= last);
                consider(q2, __float_as_uint(q2.w), i2 <= last);
                consider(q3, __float_as_uint(q3.w), i3 <= last);
            } else {
                const float4 q0 = grid.pts[s0], q1 = grid.pts[s1], q2 = grid.pts[s2], q3 = grid.pts[s3];
                consider(q0, s0, true);
                consider(q1, s1, i1 <= last);
                consider(q2, s2, i2 <= last);
                consider(q3, s3, i3 <= last);
            }
            const unsigned nx = idx + 4;
            k += (E0 <= nx ? 1u : 0u) + (E1 <= nx ? 1u : 0u) + (E2 <= nx ? 1u : 0u) + (E3 <= nx ? 1u : 0u);
        }
    }
    // ---- merge and output: as in ivox_knn_kernel (G == 4: lane 0 writes all five)
    const double m0 = group_min_dkey<G>(t5[0]);
    if (dkey_valid(m0)) {
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double m = j == 0 ? m0 : group_min_dkey<G>(t5[0]);
            const bool mv = dkey_valid(m);
            cnt += mv ? 1 : 0;
            if (mv && __double_as_longlong(t5[0]) == __double_as_longlong(m)) {
                t5[0] = t5[1]; t5[1] = t5[2]; t5[2] = t5[3]; t5[3] = t5[4]; t5[4] = kNone;
            }
            if (sub == 0 && active) store_nn_row(&nn_pts[(size_t)q * 5 + j], mv ? grid.pts[dkey_slot(m)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1)));
        }
        if (sub == 0 && active) nn_cnt[q] = (unsigned char)cnt;
    }
}

}  // namespace fls
