// kernels_ivox_coop.hpp -- the production iVox point-to-plane path, split for the machine:
//
//   ivox_knn_kernel<G>   G lanes cooperate on ONE source point: the 19 voxel probes are dealt round-robin to the G
//                        lanes (all probe loads of a wave are in flight together); the hit voxels' points are then
//                        split into equal contiguous ranges per lane (BAL: per-group LDS voxel table) or taken
//                        voxel-wise (hash-table fallback, G = 8); every lane scans its candidates four loads at a
//                        time into a private sorted top-5 of keys {float-bits(d2) : map slot} held as IEEE doubles
//                        (v_min_f64 / v_max_f64 insertion), and the G private lists are merged by five rounds of a
//                        DPP group-min.  58 VGPRs -> 8 waves/SIMD, G x more waves than points/64: the dependent
//                        gathers (cell -> voxel points) are hidden by thread-level parallelism instead of being
//                        serialised in one lane (the first, fused kernel spent 117 us per launch that way).
//                        Output: nearest_points_[i] (<=5 float4 {x,y,z,id}) + count; untouched when no
//                        candidate exists (ivox_map.cpp:21-23 quirk).
//   p2plane_fit_solve_kernel  one lane per source point: 5x3 column-pivoted Householder plane fit, gates,
//                        Jacobian (FP64), the Q1 stale-slot rule, DPP wave reduction of the 6x6 system; the last
//                        workgroup to finish also runs the Gauss-Newton tail (solve, pose update, stop rule).
//
// Exact-tie rule of the selection: lower map slot wins (the reference's order under exact float ties is
// libstdc++-introselect-defined; parity tests count such queries).
#pragma once
#include "kernels_p2plane.hpp"

namespace fls {

// NEARBY18 offsets (ivox_map.cpp:50-54) packed 2 bits per entry (value + 1), entry k at bits [2k, 2k+1]:
// a per-lane probe index needs no table load.
//   dx: 0,-1,1,0,0,0,0,1,-1,1,-1,1,-1,1,-1,0,0,0,0
//   dy: 0,0,0,1,-1,0,0,1,1,-1,-1,0,0,0,0,1,-1,1,-1
//   dz: 0,0,0,0,0,-1,1,0,0,0,0,1,1,-1,-1,1,1,-1,-1
__device__ __forceinline__ void nearby18(const int k, int& dx, int& dy, int& dz) {
    constexpr unsigned long long DX = 0x1548889561ull, DY = 0x895429495ull, DZ = 0x282956155ull;
    dx = (int)((DX >> (2 * k)) & 3ull) - 1;
    dy = (int)((DY >> (2 * k)) & 3ull) - 1;
    dz = (int)((DZ >> (2 * k)) & 3ull) - 1;
}

// Butterfly exchange partners without LDS: lane^1 and lane^2 are quad permutes, the third pairing uses
// row_half_mirror (lane i <-> 7-i inside each group of 8) -- any perfect pairing works for a min / sum.
template <int STEP>
__device__ __forceinline__ unsigned dpp_pair_u32(const unsigned v) {
    static_assert(STEP >= 0 && STEP <= 3, "");
    if (STEP == 0) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    if (STEP == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    if (STEP == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);                 // row_mirror (16 lanes)
}
template <int G>
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v) {
#define FLS_MIN_STEP(S)                                                                               \
    {                                                                                                 \
        const unsigned lo = dpp_pair_u32<S>((unsigned)(v & 0xffffffffull));                          \
        const unsigned hi = dpp_pair_u32<S>((unsigned)(v >> 32));                                    \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                            \
        v = o < v ? o : v;                                                                            \
    }
    FLS_MIN_STEP(0)
    if (G >= 4) FLS_MIN_STEP(1)
    if (G >= 8) FLS_MIN_STEP(2)
    if (G >= 16) FLS_MIN_STEP(3)
#undef FLS_MIN_STEP
    if (G == 32) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), 16, 64);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), 16, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}
template <int G>
__device__ __forceinline__ int group_sum_i32(int v) {
    v += (int)dpp_pair_u32<0>((unsigned)v);
    if (G >= 4) v += (int)dpp_pair_u32<1>((unsigned)v);
    if (G >= 8) v += (int)dpp_pair_u32<2>((unsigned)v);
    if (G >= 16) v += (int)dpp_pair_u32<3>((unsigned)v);
    if (G == 32) v += __shfl_xor(v, 16, 64);
    return v;
}

// Selection keys as IEEE doubles: the 64-bit key {float-bits(d2) + kKeyBias : map slot} read as a positive NORMAL
// double orders exactly like the unsigned integer (sign 0, exponent field >= 1 thanks to the bias, never Inf/NaN
// because d2 < 25), so a sorted insertion is nine full-rate v_min_f64 / v_max_f64 instead of five 64-bit
// compares and twenty selects.  "No candidate" = high word kKeyNoneHi (largest finite exponent; any low word).
constexpr unsigned kKeyBias = 0x00100000u, kKeyNoneHi = 0x7FEFFFFFu, kKeyValidLimit = 0x7FE00000u;
__device__ __forceinline__ double kmin_f64(const double a, const double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double kmax_f64(const double a, const double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double make_dkey(const float d2, const unsigned slot, const bool ok) {
    return __hiloint2double((int)(ok ? __float_as_uint(d2) + kKeyBias : kKeyNoneHi), (int)slot);
}
__device__ __forceinline__ bool dkey_valid(const double k) { return (unsigned)__double2hiint(k) < kKeyValidLimit; }
__device__ __forceinline__ unsigned dkey_slot(const double k) { return (unsigned)__double2loint(k); }
__device__ __forceinline__ void top5_insert_dkey(double (&t)[5], const double key) {
    double x = kmin_f64(t[4], key);
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        const double lo = kmin_f64(t[j], x), hi = kmax_f64(t[j], x);
        t[j + 1] = hi;
        x = lo;
    }
    t[0] = x;
}
template <int G>
__device__ __forceinline__ double group_min_dkey(double v) {
#define FLS_DMIN_STEP(S)                                                                              \
    {                                                                                                 \
        const unsigned lo = dpp_pair_u32<S>((unsigned)__double2loint(v));                             \
        const unsigned hi = dpp_pair_u32<S>((unsigned)__double2hiint(v));                             \
        v = kmin_f64(v, __hiloint2double((int)hi, (int)lo));                                          \
    }
    FLS_DMIN_STEP(0)
    if (G >= 4) FLS_DMIN_STEP(1)
    if (G >= 8) FLS_DMIN_STEP(2)
#undef FLS_DMIN_STEP
    return v;
}

// candidate index -> map slot as a compare / select chain on VALUES (written as a function of scalars: a lambda
// capturing the offsets by reference made the compiler select between ADDRESSES and load through them)
template <int R>
__device__ __forceinline__ unsigned slot_select(const unsigned idx, const unsigned p1, const unsigned p2, const unsigned p3, const unsigned p4,
                                                const unsigned o0, const unsigned o1, const unsigned o2, const unsigned o3, const unsigned o4) {
    unsigned sel = R == 5 ? o4 : R == 4 ? o3 : R == 3 ? o2 : R == 2 ? o1 : o0;
    if (R > 4) sel = idx < p4 ? o3 : sel;
    if (R > 3) sel = idx < p3 ? o2 : sel;
    if (R > 2) sel = idx < p2 ? o1 : sel;
    if (R > 1) sel = idx < p1 ? o0 : sel;
    return idx + sel;
}

// nearest_points_ rows: 9.2 MB per launch at the benchmark size, flushed from the write-back L2 at the kernel boundary
// (2.7 us of this kernel + 3.2 us of re-reads in the fit kernel, measured by removing the stores).  -DFLS_NN_STORE=1 sends
// them out as nontemporal stores instead: measured -0.5 us per launch, -0.5 us per Match (inside the noise), so the plain
// store stays the default; write-through (sc1) stores were slower (18.6-21.2 us per launch).
#ifndef FLS_NN_STORE
#define FLS_NN_STORE 0
#endif
#ifndef FLS_KNN_PIPE
#define FLS_KNN_PIPE 0
#endif
typedef float fls_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nn_row(float4* p, const float4 v) {
#if FLS_NN_STORE == 0
    *p = v;
#else
    fls_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<fls_v4f*>(p));
#endif
}

template <int G, bool COUNT, bool DENSE, bool FIRST, bool BAL = false>
__global__ void __launch_bounds__(256)
ivox_knn_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                const GnState* __restrict__ st, const Pose16 T0, const DevGrid grid, const BrickDir bd,
                const float inv_res, float4* __restrict__ nn_pts /* [n][5] */, unsigned char* __restrict__ nn_cnt,
                unsigned char* __restrict__ flag, TrafficCounters* __restrict__ tc, const int chunk,
                unsigned* __restrict__ nn_ids /* [n][8]: map slots of the neighbours (ids form); nullptr: rows form */,
                const int nn_prev /* FIRST: size of nearest_points_ before this Match; the grown tail starts empty (:257 resize) */,
                float* __restrict__ dev_copy /* FIRST, may be null: sx / sy / sz point into the pinned staging buffer (host memory);
                                                leave the device copy x[n] | y[n] | z[n] here for the launches that follow */) {
    static_assert(G == 4 || G == 8, "group size");
    constexpr int QPB = 256 / G;       // queries per workgroup
    constexpr int R = (19 + G - 1) / G;  // probe rounds per lane
    // XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with its own
    // 4 MiB L2.  Re-map so that XCD x walks CHUNKS of `chunk` consecutive workgroups (chunk * QPB consecutive points of
    // the ring-major, hence spatially coherent, scan), the chunks of the 8 XCDs interleaved.  One contiguous eighth of
    // the scan per XCD (the first version) caches best but balances worst: the rings differ in map density, the
    // busiest XCD carried 38 % more candidate work than the mean.  Affects speed only.
    // (the grid is launched rounded up to a multiple of 8 * chunk so that the re-map is a bijection)
    const int nb = (n + QPB - 1) / QPB;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int lb = ((seq / chunk) * 8 + xcd) * chunk + (seq % chunk);
    const int sub = threadIdx.x % G;
    const int q = lb * QPB + threadIdx.x / G;
    const bool active = lb < nb && q < n;
    // issue the state / source loads before looking at the stop flag: one memory round trip instead of three
    // (source loads are unconditional on a clamped index: a guarded load costs a branch and a wait each).
    // The FIRST iteration of a Match takes the initial pose from the launch arguments and ignores the
    // (stale) device state; it also clears the per-point valid flags (std::fill once per Match, Q1).
    const int done = FIRST ? 0 : st->done;
    double T[12];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) T[c * 3 + r] = FIRST ? T0.m[c * 4 + r] : st->T[c * 4 + r];
    const int qq = active ? q : 0;
    const float px = sx[qq], py = sy[qq], pz = sz[qq];
    if (done) return;
    if (FIRST && active && sub == 0) {
        if (dev_copy) { dev_copy[q] = px; dev_copy[(size_t)n + q] = py; dev_copy[2 * (size_t)n + q] = pz; }
        flag[q] = 0;
        if (q >= nn_prev) nn_cnt[q] = 0;  // (an unaligned hipMemsetAsync of the tail cost up to three fill kernels in front of the Match)
    }
    const double x = px, y = py, z = pz;
    const float ptx = (float)(((T[0] * x + T[3] * y) + T[6] * z) + T[9]);
    const float pty = (float)(((T[1] * x + T[4] * y) + T[7] * z) + T[10]);
    const float ptz = (float)(((T[2] * x + T[5] * y) + T[8] * z) + T[11]);
    const float fx = roundf(ptx * inv_res), fy = roundf(pty * inv_res), fz = roundf(ptz * inv_res);
    const bool in_range = active && fabsf(fx) < (float)kKeyLimit && fabsf(fy) < (float)kKeyLimit && fabsf(fz) < (float)kKeyLimit;
    const int kx = in_range ? (int)fx : 0, ky = in_range ? (int)fy : 0, kz = in_range ? (int)fz : 0;

    // phase 1: this lane's probes (k = sub, sub+G, ...): issue every first-slot load before resolving any.
    // Written as explicit per-round scalars (no indexed arrays) so that nothing lands in scratch.
    const double kNone = __hiloint2double((int)kKeyNoneHi, -1);
    double t5[5] = {kNone, kNone, kNone, kNone, kNone};
    unsigned long long c_hits = 0, c_cand = 0, c_probes = 0;
    auto probe_key = [&](const int r, bool& pv) -> unsigned long long {
        const int k = sub + G * r;
        pv = in_range && k < 19;
        int ox, oy, oz;
        nearby18(k < 19 ? k : 0, ox, oy, oz);
        return pack_key(kx + ox, ky + oy, kz + oz);
    };
    auto first_load = [&](const bool pv, const unsigned long long key, unsigned& h) -> HashEntry {
        h = hash_key(key) & grid.mask;
        return pv ? grid.table[h] : HashEntry{kEmptyKey, 0u, 0u};
    };
    // resolve a probe (linear probing on collision) to its voxel's point range; count = 0 on a miss
    auto resolve = [&](const bool pv, const unsigned long long key, unsigned h, HashEntry ek, unsigned& beg, unsigned& cnt) {
        while (pv && ek.key != key && ek.key != kEmptyKey) {
            h = (h + 1) & grid.mask;
            ek = grid.table[h];
        }
        const bool hit = pv && ek.key == key;
        beg = hit ? ek.begin : 0u;
        cnt = hit ? ek.count : 0u;
        if (COUNT && pv) { c_probes++; if (hit) { c_hits++; c_cand += ek.count; } }
    };
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (DENSE) {
        // brick image (device_common.hpp): ONE directory look-up per query -- the same 16-byte entry for the lanes of a group -- then
        // every probe is a slab offset from the query's own cell (the slab's halo mirrors the neighbouring bricks' boundary cells):
        // no bounds checks, one aligned 8-byte load per probe, neighbouring voxels share cache lines
        const int bx = kx >> kBrickLog, by = ky >> kBrickLog, bz = kz >> kBrickLog;
        const unsigned long long bkey = pack_key(bx, by, bz);
        unsigned h = brick_hash(bx, by, bz) & bd.mask;
        HashEntry be = bd.table[h];
        while (be.key != bkey && be.key != kEmptyKey) {  // (linear probing; the directory is at most a quarter full)
            h = (h + 1) & bd.mask;
            be = bd.table[h];
        }
        const bool have = in_range && be.key == bkey && be.begin < bd.n_cap;
        const unsigned cbase = (have ? be.begin : 0u) * kBrickStride +
                               brick_slab_index((kx & (kBrickSide - 1)) + 1, (ky & (kBrickSide - 1)) + 1, (kz & (kBrickSide - 1)) + 1);
        auto cell = [&](const int r, unsigned& beg, unsigned& cnt) {
            const int k = sub + G * r;
            int ox, oy, oz;
            nearby18(k < 19 ? k : 0, ox, oy, oz);
            const bool ok = have && k < 19;
            const uint2 e = bd.cells[cbase + (unsigned)((oz * kBrickStored + oy) * kBrickStored + ox)];  // (always inside the slab)
            beg = e.x;
            cnt = ok ? e.y : 0u;
            if (COUNT && in_range && k < 19) { c_probes++; if (cnt) { c_hits++; c_cand += cnt; } }
        };
        cell(0, b0, c0);
        if (R > 1) cell(1, b1, c1);
        if (R > 2) cell(2, b2, c2);
        if (R > 3) cell(3, b3, c3);
        if (R > 4) cell(4, b4, c4);
    } else {
        bool pv0 = false, pv1 = false, pv2 = false, pv3 = false, pv4 = false;
        unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0;
        unsigned h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
        HashEntry e0{kEmptyKey, 0u, 0u}, e1 = e0, e2 = e0, e3 = e0, e4 = e0;
        k0 = probe_key(0, pv0); e0 = first_load(pv0, k0, h0);
        if (R > 1) { k1 = probe_key(1, pv1); e1 = first_load(pv1, k1, h1); }
        if (R > 2) { k2 = probe_key(2, pv2); e2 = first_load(pv2, k2, h2); }
        if (R > 3) { k3 = probe_key(3, pv3); e3 = first_load(pv3, k3, h3); }
        if (R > 4) { k4 = probe_key(4, pv4); e4 = first_load(pv4, k4, h4); }
        resolve(pv0, k0, h0, e0, b0, c0);
        if (R > 1) resolve(pv1, k1, h1, e1, b1, c1);
        if (R > 2) resolve(pv2, k2, h2, e2, b2, c2);
        if (R > 3) resolve(pv3, k3, h3, e3, b3, c3);
        if (R > 4) resolve(pv4, k4, h4, e4, b4, c4);
    }
    // scan this lane's <= R voxels, 4 point loads in flight per trip (the dependent-load latency is paid
    // once per four candidates; the tail of a voxel re-reads its last point, masked out)
    auto consider = [&](const float4 p, const unsigned s, const bool ok) {
        const float dx = p.x - ptx, dy = p.y - pty, dz = p.z - ptz;
        const float d2 = dx * dx + (dy * dy + dz * dz);  // Eigen Vector3f::squaredNorm order
        top5_insert_dkey(t5, make_dkey(d2, s, ok && d2 < 25.0f));  // max_range 5.0 squared (never binds at 0.5 m voxels)
    };
    // one flattened loop over this lane's <= R voxels: trip count = ceil(lane total / 4), not the sum of
    // per-voxel maxima
    const unsigned p1 = c0, p2 = p1 + c1, p3 = p2 + c2, p4 = p3 + c3, tot = p4 + c4;
    // candidate index -> map slot, branch-free: slot = idx + (begin - prefix) of the voxel the index falls in
    // (unsigned wrap-around is intended)
    const unsigned o0 = b0, o1 = b1 - p1, o2 = b2 - p2, o3 = b3 - p3, o4 = b4 - p4;
    auto slot_of = [=](const unsigned idx) -> unsigned { return slot_select<R>(idx, p1, p2, p3, p4, o0, o1, o2, o3, o4); };
    if (BAL && G == 4) {
        // Balanced split: the group's candidates (all hit voxels of the query, concatenated) are cut into four equal
        // contiguous ranges, one per lane, instead of whole voxels per lane: a wave runs max-over-lanes trips of the
        // loop below, and voxels hold 1..20+ points (4.7 trips of four candidates per wave before, 2.8 after).
        // The hit voxels are compacted into a per-group LDS table {prefix end, begin - prefix start}; a lane walks
        // its range through a 5-entry window of that table (four candidates span at most five voxels).
        __shared__ __attribute__((aligned(16))) unsigned s_end[QPB][24];
        __shared__ __attribute__((aligned(16))) unsigned s_off[QPB][24];
        const int g = threadIdx.x / G;
        const unsigned nz = (c0 ? 1u : 0u) + (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u) + (c4 ? 1u : 0u);
        const unsigned packed = tot * 32u + nz;  // candidates (< 2^27) and hit voxels (<= 19 per group) of this lane
        const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x00, 0xf, 0xf, true);  // quad broadcasts
        const unsigned t1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x55, 0xf, 0xf, true);
        const unsigned t2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xAA, 0xf, 0xf, true);
        const unsigned t3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xFF, 0xf, 0xf, true);
        const unsigned before = (sub > 0 ? t0 : 0u) + (sub > 1 ? t1 : 0u) + (sub > 2 ? t2 : 0u), all = t0 + t1 + t2 + t3;
        const unsigned TOT = all >> 5;
        unsigned pos = before & 31u, run = before >> 5;
        // every entry behind the last real one is a sentinel (the window looks four entries ahead, the start search
        // reads twenty): the whole row is filled first, the real entries overwrite (LDS operations of a wave are in order)
        {
            uint2* const rowp = reinterpret_cast<uint2*>(&s_end[g][6 * sub]);
            rowp[0] = make_uint2(~0u, ~0u); rowp[1] = make_uint2(~0u, ~0u); rowp[2] = make_uint2(~0u, ~0u);
        }
        if (c0) { s_off[g][pos] = b0 - run; run += c0; s_end[g][pos] = run; ++pos; }
        if (c1) { s_off[g][pos] = b1 - run; run += c1; s_end[g][pos] = run; ++pos; }
        if (c2) { s_off[g][pos] = b2 - run; run += c2; s_end[g][pos] = run; ++pos; }
        if (c3) { s_off[g][pos] = b3 - run; run += c3; s_end[g][pos] = run; ++pos; }
        if (c4) { s_off[g][pos] = b4 - run; run += c4; s_end[g][pos] = run; ++pos; }
        // a group lives inside one wave and the LDS serves a wave's operations in order: no workgroup barrier, only a
        // compiler-level ordering point between the table writes and the cross-lane reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned Q = (TOT + 3u) >> 2, a = sub * Q, e = a + Q < TOT ? a + Q : TOT;
        unsigned k = 0;  // first table entry whose range reaches past a
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const uint4 e4 = *reinterpret_cast<const uint4*>(&s_end[g][4 * v]);
            k += (e4.x <= a ? 1u : 0u) + (e4.y <= a ? 1u : 0u) + (e4.z <= a ? 1u : 0u) + (e4.w <= a ? 1u : 0u);
        }
#if FLS_KNN_PIPE
        // software-pipelined variant (A/B switch): the four loads of trip t + 1 are issued before the candidates of trip t are
        // consumed, so a lane has up to eight points in flight
        {
            const unsigned last = e - 1;
            unsigned cs0 = 0, cs1 = 0, cs2 = 0, cs3 = 0;
            float4 cq0 = make_float4(0.f, 0.f, 0.f, 0.f), cq1 = cq0, cq2 = cq0, cq3 = cq0;
            auto fetch = [&](const unsigned idx, unsigned& s0, unsigned& s1, unsigned& s2, unsigned& s3, float4& q0, float4& q1, float4& q2, float4& q3) {
                const unsigned E0 = s_end[g][k], E1 = s_end[g][k + 1], E2 = s_end[g][k + 2], E3 = s_end[g][k + 3];
                const unsigned O0 = s_off[g][k], O1 = s_off[g][k + 1], O2 = s_off[g][k + 2], O3 = s_off[g][k + 3], O4 = s_off[g][k + 4];
                const unsigned i1 = idx + 1, i2 = idx + 2, i3 = idx + 3;
                s0 = slot_select<5>(idx, E0, E1, E2, E3, O0, O1, O2, O3, O4);
                s1 = slot_select<5>(i1 < last ? i1 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
                s2 = slot_select<5>(i2 < last ? i2 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
                s3 = slot_select<5>(i3 < last ? i3 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
                q0 = grid.pts[s0]; q1 = grid.pts[s1]; q2 = grid.pts[s2]; q3 = grid.pts[s3];
                const unsigned nx = idx + 4;
                k += (E0 <= nx ? 1u : 0u) + (E1 <= nx ? 1u : 0u) + (E2 <= nx ? 1u : 0u) + (E3 <= nx ? 1u : 0u);
            };
            if (a < e) fetch(a, cs0, cs1, cs2, cs3, cq0, cq1, cq2, cq3);
            for (unsigned idx = a; idx < e; idx += 4) {
                unsigned ns0 = 0, ns1 = 0, ns2 = 0, ns3 = 0;
                float4 nq0 = cq0, nq1 = cq0, nq2 = cq0, nq3 = cq0;
                if (idx + 4 < e) fetch(idx + 4, ns0, ns1, ns2, ns3, nq0, nq1, nq2, nq3);
                consider(cq0, cs0, true);
                consider(cq1, cs1, idx + 1 <= last);
                consider(cq2, cs2, idx + 2 <= last);
                consider(cq3, cs3, idx + 3 <= last);
                cs0 = ns0; cs1 = ns1; cs2 = ns2; cs3 = ns3;
                cq0 = nq0; cq1 = nq1; cq2 = nq2; cq3 = nq3;
            }
        }
#else
        for (unsigned idx = a; idx < e; idx += 4) {
            const unsigned E0 = s_end[g][k], E1 = s_end[g][k + 1], E2 = s_end[g][k + 2], E3 = s_end[g][k + 3];
            const unsigned O0 = s_off[g][k], O1 = s_off[g][k + 1], O2 = s_off[g][k + 2], O3 = s_off[g][k + 3], O4 = s_off[g][k + 4];
            const unsigned last = e - 1;
            const unsigned i1 = idx + 1, i2 = idx + 2, i3 = idx + 3;
            const unsigned s0 = slot_select<5>(idx, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s1 = slot_select<5>(i1 < last ? i1 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s2 = slot_select<5>(i2 < last ? i2 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s3 = slot_select<5>(i3 < last ? i3 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const float4 q0 = grid.pts[s0], q1 = grid.pts[s1], q2 = grid.pts[s2], q3 = grid.pts[s3];
            consider(q0, s0, true);
            consider(q1, s1, i1 <= last);
            consider(q2, s2, i2 <= last);
            consider(q3, s3, i3 <= last);
            const unsigned nx = idx + 4;
            k += (E0 <= nx ? 1u : 0u) + (E1 <= nx ? 1u : 0u) + (E2 <= nx ? 1u : 0u) + (E3 <= nx ? 1u : 0u);
        }
#endif
    } else
    for (unsigned j = 0; j < tot; j += 4) {
        const unsigned last = tot - 1;
        const unsigned i1 = j + 1, i2 = j + 2, i3 = j + 3;
        const unsigned s0 = slot_of(j), s1 = slot_of(i1 < last ? i1 : last), s2 = slot_of(i2 < last ? i2 : last),
                       s3 = slot_of(i3 < last ? i3 : last);
        const float4 p0 = grid.pts[s0], p1 = grid.pts[s1], p2 = grid.pts[s2], p3 = grid.pts[s3];
        consider(p0, s0, true);
        consider(p1, s1, i1 <= last);
        consider(p2, s2, i2 <= last);
        consider(p3, s3, i3 <= last);
    }
    // phase 2: merge the G private lists: five rounds of group-min + pop (keys are unique -- the slot is the low
    // word -- so exactly one lane pops per round); the neighbour count is the number of valid minima
    const double m0 = group_min_dkey<G>(t5[0]);
    if (dkey_valid(m0)) {  // uniform within the group: at least one candidate (else nothing is written, ivox_map.cpp:21-23)
        int cnt = 0;
        double mine = kNone;  // G == 8: the j-th smallest key lands in lane sub == j
        unsigned sid[5] = {~0u, ~0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double m = j == 0 ? m0 : group_min_dkey<G>(t5[0]);
            const bool mv = dkey_valid(m);
            cnt += mv ? 1 : 0;
            if (mv && __double_as_longlong(t5[0]) == __double_as_longlong(m)) {
                t5[0] = t5[1]; t5[1] = t5[2]; t5[2] = t5[3]; t5[3] = t5[4]; t5[4] = kNone;
            }
            if (G >= 8) { if (sub == j) mine = m; }
            else if (nn_ids) sid[j] = mv ? dkey_slot(m) : ~0u;
            else if (sub == 0 && active) {  // G == 4, rows form: lane 0 writes all five
                store_nn_row(&nn_pts[(size_t)q * 5 + j], mv ? grid.pts[dkey_slot(m)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1)));
            }
        }
        // Neighbour lists in IDS form (round 3): 20 bytes of map slots per point instead of 80 bytes of gathered rows -- the rows were the
        // path's only real HBM traffic (9.2 MB written here and re-read by the fit kernel every iteration); the fit kernel gathers
        // the five points from the (cache-resident) map image instead.  Bit 7 of the count byte says which form a point's list has
        // (0x80 = rows): a point without candidates keeps its previous list in whatever form it was (Q15); the lists are turned into
        // rows before anything moves map slots (matcher_p2plane_ivox.hpp::ensure_nn_rows).
        if (nn_ids) {
            if (G >= 8) { if (sub < 5 && active) nn_ids[(size_t)q * 8 + sub] = dkey_valid(mine) ? dkey_slot(mine) : ~0u; }
            else if (sub == 0 && active) {
                *reinterpret_cast<uint4*>(nn_ids + (size_t)q * 8) = make_uint4(sid[0], sid[1], sid[2], sid[3]);
                nn_ids[(size_t)q * 8 + 4] = sid[4];
            }
            if (sub == 0 && active) nn_cnt[q] = (unsigned char)cnt;
        } else {
            if (G >= 8 && sub < 5 && active)
                nn_pts[(size_t)q * 5 + sub] = dkey_valid(mine) ? grid.pts[dkey_slot(mine)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (sub == 0 && active) nn_cnt[q] = (unsigned char)(cnt | 0x80);
        }
    }
    if (COUNT) {
        const double p = wave_sum_u((double)c_probes), hsum = wave_sum_u((double)c_hits), c = wave_sum_u((double)c_cand);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&tc->probes, (unsigned long long)p);
            atomicAdd(&tc->hits, (unsigned long long)hsum);
            atomicAdd(&tc->cand, (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// p2plane_fit_solve_kernel: fit + residual + block reduction (one lane per source point, reading what
// ivox_knn_kernel left behind), and the LAST workgroup to finish runs the Gauss-Newton
// tail (reduce the block partials, 6x6 solve, pose update, stop rule) -- one launch per iteration fewer.
// Cross-workgroup hand-off (placement independent, no spinning): every workgroup publishes its partial row with
// write-through (sc1) relaxed agent-scope stores, the storing wave drains (s_waitcnt vmcnt(0)), __syncthreads, one
// lane takes a ticket with an agent-scope atomic; the workgroup that draws the last ticket reads all rows with sc1
// (relaxed agent-scope) loads.  No release fence: an agent-scope release writes back the XCD's whole L2, which
// this kernel has just dirtied with 28 KB of Jacobian rows per workgroup (measured: 7 us on the critical path).
// The ticket counters (kTicketWords unsigned words) are reset by their last arrivers, so they are zero at every launch.
// A point's Jacobian row and flag go to memory AFTER the workgroup has handed its partial row over (in the workgroup that runs the tail: after
// the tail): __syncthreads() waits for every store a wave has issued, and 28 KB of rows per workgroup in front of the hand-off's barriers sat on
// the last workgroup's critical path (round 5: -0.3 us per launch here, -1.7 us per Match on a 5.8 k-point scan; same registers, same bits:
// profiles/r05_ll_ab_late_stores_x_fanin.log).  Nothing inside the launch reads them back.
// Where the time of the tail workgroup goes (profiles/r05_ll_fanin_stamps.log, -DFLS_TIMING, ~2.3 ticks per ns): its own fit 4.2 us, waiting for
// the slowest of its eight waves + row store + drain 3.2 us, two ticket levels 1.0 us, row read + reduction 1.6 us, LDL^T 1.45 us, pose update
// + state + mailbox 1.9 us.
// ---------------------------------------------------------------------------------------------
constexpr int kFitThreads = 512;
template <bool FIRST, int NT = kFitThreads>
__global__ void __launch_bounds__(NT)
p2plane_fit_solve_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                         GnState* __restrict__ st, const Pose16 T0, const float4* __restrict__ nn_pts,
                         const unsigned char* __restrict__ nn_cnt, double* __restrict__ Jst /* [7][n] */, unsigned char* __restrict__ flag,
                         double* __restrict__ partials, unsigned* __restrict__ ticket, Mailbox* __restrict__ mb, const unsigned match_id,
                         const double plane_thres, const double rot_thr, const double pos_thr, const int shards,
                         const unsigned* __restrict__ nn_ids /* may be null */, const float4* __restrict__ map_pts, const unsigned n_slots) {
    const int i = blockIdx.x * NT + threadIdx.x;
    const int done = FIRST ? 0 : st->done;
    double T44[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T44[k] = FIRST ? T0.m[k] : st->T[k];
    const double last_rot = FIRST ? 0.0 : st->last_rot, last_pos = FIRST ? 0.0 : st->last_pos;
    const int it = FIRST ? 0 : st->iter;
    // every per-point input is loaded up front on a clamped index (one memory round trip; the neighbour points
    // are fetched whether or not all five exist, the stale flag whether or not it is needed)
    const int ii = i < n ? i : 0;
    const int cb = nn_cnt[ii];
    const int cnt = cb & 7;
    float4 nn[5];
    if (nn_ids) {  // uniform: lists in ids form unless a point's count byte says rows (bit 7)
        const uint4 i4 = *reinterpret_cast<const uint4*>(nn_ids + (size_t)ii * 8);
        const unsigned i5 = nn_ids[(size_t)ii * 8 + 4];
        const unsigned sl[5] = {i4.x, i4.y, i4.z, i4.w, i5};
        const bool rows_form = (cb & 0x80) != 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float4* src = rows_form ? nn_pts + (size_t)ii * 5 + j : map_pts + (sl[j] < n_slots ? sl[j] : 0u);
            nn[j] = *src;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) nn[j] = nn_pts[(size_t)ii * 5 + j];
    }
    const float px = sx[ii], py = sy[ii], pz = sz[ii];
    const unsigned char stale = FIRST ? (unsigned char)0 : flag[ii];  // the first kNN launch of a Match cleared the flags
    if (done) return;
    __shared__ LoamTailSmem sm;
    __shared__ double wsum[NT / 64][32];
    __shared__ unsigned s_ticket;
#ifdef FLS_TIMING
    const long long t_begin = (long long)__builtin_readcyclecounter();
#endif
    bool contrib = false, fresh = false;
    double J[6] = {0, 0, 0, 0, 0, 0}, res = 0.0;
    auto store_point = [&]() {  // (called once, after the hand-off)
        if (!fresh) return;
#pragma unroll
        for (int a = 0; a < 6; ++a) Jst[(size_t)a * n + i] = J[a];
        Jst[(size_t)6 * n + i] = res;
        flag[i] = 1;
    };
    if (i < n) {
        bool valid_now = false;
        if (cnt == 5) {
            const double x = px, y = py, z = pz;
            const float ptx = (float)(((T44[0] * x + T44[4] * y) + T44[8] * z) + T44[12]);
            const float pty = (float)(((T44[1] * x + T44[5] * y) + T44[9] * z) + T44[13]);
            const float ptz = (float)(((T44[2] * x + T44[6] * y) + T44[10] * z) + T44[14]);
            valid_now = plane_residual_dev(nn, px, py, pz, ptx, pty, ptz, T44, plane_thres, J, res);
        }
        if (valid_now) {
            fresh = true;
            contrib = true;
        } else if (stale) {  // Q1: stale contribution of an earlier iteration
#pragma unroll
            for (int a = 0; a < 6; ++a) J[a] = Jst[(size_t)a * n + i];
            res = Jst[(size_t)6 * n + i];
            contrib = true;
        }
    }
#ifdef FLS_TIMING
    const long long t_fit = (long long)__builtin_readcyclecounter();
#endif
    // wave sums -> LDS -> one row per workgroup (fixed order: wave 0 + wave 1 + ...)
#if FLS_FIT_MFMA
    __shared__ __attribute__((aligned(64))) double mfma_tile[NT / 64][512];
    reduce_rank1_mfma_and_store(contrib, J, res, &wsum[threadIdx.x >> 6][0], &mfma_tile[threadIdx.x >> 6][0]);
#else
    reduce_rank1_and_store(contrib, J, res, &wsum[threadIdx.x >> 6][0]);
#endif
#ifdef FLS_TIMING
    const long long t_red = (long long)__builtin_readcyclecounter();
#endif
    __syncthreads();
    if (threadIdx.x < 29) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += wsum[w][threadIdx.x];
        // write-through (sc1) publish: visible to every XCD once this wave's stores have drained -- no L2 write-back
        __hip_atomic_store((unsigned long long*)partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x,
                           (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
#ifdef FLS_TIMING
    const long long t_drain = (long long)__builtin_readcyclecounter();
#endif
    __syncthreads();
    // sharded fan-in (kernels_p2plane.hpp::fanin_last_arriver): which workgroup arrives last
    if (threadIdx.x == 0) s_ticket = fanin_last_arriver(ticket, shards);
    __syncthreads();
    if (!s_ticket) { store_point(); return; }
    // ---- last workgroup: Gauss-Newton tail (reads the rows with sc1 loads: no acquire fence either) ----
#ifdef FLS_TIMING
    if (threadIdx.x == 0) { st->dbg[0] = t_begin; st->dbg[13] = t_fit; st->dbg[14] = t_red; st->dbg[15] = t_drain; }
#endif
    FLS_STAMP(1);
    loam_tail<NT, true>(st, sm, nullptr, 0, partials, (int)gridDim.x, rot_thr, pos_thr, T44, last_rot, last_pos, it, mb, match_id);
    store_point();  // (every thread comes back from the tail: waves 1.. at once, wave 0 after it has published)
}

// ---------------------------------------------------------------------------------------------
// Map maintenance on the device
//   ivox_apply_updates_kernel   scatter {slot, point} and {cell, begin, count} records into the resident image
//   ivox_add_decide_kernel      the per-point down-sampling decision of LoamPointToPlaneIVOX::AddCloudToLocalMap
//                               (loam_point_to_plane_ivox.h:89-128) on the neighbour lists left by the last
//                               PlanerMatch: code 0 = drop, 1 = points_to_add, 2 = point_no_need_downsample;
//                               pw = pcl::transformPoint(p, T_) (double -> float)
// ---------------------------------------------------------------------------------------------
struct PtUpdDev { unsigned slot; float x, y, z; int id; };
struct CellUpdDev { unsigned long long idx; unsigned begin, count; };

// An image that arrived from another process (fls_map_image_import): every {begin, count} must stay inside the point array and every
// directory entry must name a brick the image holds, or a query kernel would read out of bounds.  bad = number of offending entries.
__global__ void __launch_bounds__(256)
ivox_image_validate_kernel(const uint2* __restrict__ cells, const unsigned long long n_cells, const HashEntry* __restrict__ dir, const unsigned long long n_dir,
                           const unsigned n_bricks_live, const HashEntry* __restrict__ table, const unsigned long long n_table, const unsigned long long used,
                           unsigned* __restrict__ bad) {
    unsigned b = 0u;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n_cells; i += stride) {
        const uint2 c = cells[i];
        b += ((unsigned long long)c.x + (unsigned long long)c.y > used) ? 1u : 0u;
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n_dir; i += stride) {
        const HashEntry e = dir[i];
        b += (e.key != kEmptyKey && e.begin >= n_bricks_live) ? 1u : 0u;
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n_table; i += stride) {
        const HashEntry e = table[i];
        b += (e.key != kEmptyKey && (unsigned long long)e.begin + (unsigned long long)e.count > used) ? 1u : 0u;
    }
    if (b) atomicAdd(bad, b);
}

__global__ void __launch_bounds__(256)
ivox_apply_updates_kernel(const PtUpdDev* __restrict__ pu, const int npu, const CellUpdDev* __restrict__ cu, const int ncu,
                          float4* __restrict__ pts, uint2* __restrict__ cells) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npu) {
        const PtUpdDev u = pu[i];
        pts[u.slot] = make_float4(u.x, u.y, u.z, __int_as_float(u.id));
    }
    if (i < ncu) {
        const CellUpdDev u = cu[i];
        cells[u.idx] = make_uint2(u.begin, u.count);
    }
}

// neighbour lists: ids form -> rows form (before map slots move, and for the host-side readers)
__global__ void __launch_bounds__(256)
ivox_nn_materialize_kernel(const unsigned* __restrict__ nn_ids, unsigned char* __restrict__ nn_cnt, const int n, const float4* __restrict__ map_pts,
                           const unsigned n_slots, float4* __restrict__ nn_pts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char cb = nn_cnt[i];
    if (cb & 0x80) return;
    if (cb & 7) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned sl = nn_ids[(size_t)i * 8 + j];
            nn_pts[(size_t)i * 5 + j] = sl < n_slots ? map_pts[sl] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        }
    }
    nn_cnt[i] = (unsigned char)(cb | 0x80);
}

// MATERIALIZE (round 3): the launch also turns the ids-form lists into rows (what ivox_nn_materialize_kernel does) -- the map update that
// follows moves slots, and this kernel gathers the same five points anyway: one launch and one gather less per mapping-mode scan.
// The grid then covers max(n, nn_n) points (lists beyond the points that are decided on still become rows).
// COUNT (round 4): the launch also does what ivox_upd_count did -- block-local exclusive counts of the two insertion codes (lx) and the
// block totals (bt) for the device-side AddPoints -- and resets the batch status words: one dependent launch less in front of the update.
template <bool MATERIALIZE>
__global__ void __launch_bounds__(256)
ivox_add_decide_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                       const Pose16 Tw, float4* __restrict__ nn_pts, unsigned char* __restrict__ nn_cnt, const int nn_n,
                       const double fs /* filter_size_map_min */, unsigned char* __restrict__ code, float4* __restrict__ pw_out,
                       const unsigned* __restrict__ nn_ids /* may be null */, const float4* __restrict__ map_pts, const unsigned n_slots,
                       uint2* __restrict__ lx /* may be null: no counting */, uint2* __restrict__ bt, unsigned* __restrict__ st_status,
                       unsigned* __restrict__ st_apply, const GnState* __restrict__ gn /* non-null: SPECULATIVE launch, see below */, const int max_it) {
    // SPECULATIVE form (round 4): the host queues the decision + update chain right behind the iterations it expects the Match to need,
    // without waiting for the result (14 us of mailbox turnaround + launch latency in front of the map update).  The launch then decides
    // ON THE DEVICE whether LoamPointToPlaneIVOX::Match would reach AddCloudToLocalMap here (:197-206): the Gauss-Newton loop has ended
    // (stop rule or max_iterations) and n_valid >= 50.  If not, the whole chain marks itself skipped (kUpdSkipped = 16) and does nothing;
    // the host queues another chain behind the iterations it adds.  The pose is the device's (GnState::T), not a launch argument.
    Pose16 Tl = Tw;
    if (gn != nullptr) {
        const bool go = (gn->done != 0 || gn->iter >= max_it) && gn->n_valid >= 50;
        if (!go) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { *st_status = 16u; *st_apply = 0u; }
            return;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) Tl.m[q] = gn->T[q];
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned char c = 0;  // this point's code (0 also for the threads beyond n)
    if (i < n || (MATERIALIZE && i < nn_n)) {
        const int cb = i < nn_n ? nn_cnt[i] : 0;
        const int cnt = cb & 7;
        const bool ids_form = nn_ids != nullptr && !(cb & 0x80);
        float4 nb[5];
        if (MATERIALIZE) {
            // all five rows now (whatever the decision below needs), written back in rows form
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (ids_form) {
                    const unsigned sl = nn_ids[(size_t)i * 8 + k];
                    nb[k] = (cnt && sl < n_slots) ? map_pts[sl] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
                } else {
                    nb[k] = (i < nn_n && cnt) ? nn_pts[(size_t)i * 5 + k] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
                }
            }
            if (i < nn_n && ids_form) {
                if (cnt) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) nn_pts[(size_t)i * 5 + k] = nb[k];
                }
                nn_cnt[i] = (unsigned char)(cb | 0x80);
            }
        }
        if (i < n) {
            const double x = sx[i], y = sy[i], z = sz[i];
            const float wx = (float)(((Tl.m[0] * x + Tl.m[4] * y) + Tl.m[8] * z) + Tl.m[12]);
            const float wy = (float)(((Tl.m[1] * x + Tl.m[5] * y) + Tl.m[9] * z) + Tl.m[13]);
            const float wz = (float)(((Tl.m[2] * x + Tl.m[6] * y) + Tl.m[10] * z) + Tl.m[14]);
            pw_out[i] = make_float4(wx, wy, wz, 0.f);
            auto neighbour = [&](const int k) -> float4 {
                if (MATERIALIZE) return nb[k];
                if (!ids_form) return nn_pts[(size_t)i * 5 + k];
                const unsigned sl = nn_ids[(size_t)i * 8 + k];
                return map_pts[sl < n_slots ? sl : 0u];
            };
            c = 1;  // no neighbours: add (:126)
            if (cnt > 0) {
                const double half = 0.5 * fs;
                const double c0 = (floor((double)wx / fs) + 0.5) * fs, c1 = (floor((double)wy / fs) + 0.5) * fs, c2 = (floor((double)wz / fs) + 0.5) * fs;
                const float4 n0 = neighbour(0);
                const double d0 = (double)n0.x - c0, d1 = (double)n0.y - c1, d2 = (double)n0.z - c2;
                if (fabs(d0) > half && fabs(d1) > half && fabs(d2) > half) {
                    c = 2;  // :103-108
                } else {
                    const double e0 = (double)wx - c0, e1 = (double)wy - c1, e2 = (double)wz - c2;
                    const double dist = (e0 * e0 + e1 * e1) + e2 * e2;
                    bool need_add = true;
                    if (cnt >= 5) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            if (!need_add) break;
                            const float4 q = neighbour(k);
                            const double f0 = (double)q.x - c0, f1 = (double)q.y - c1, f2 = (double)q.z - c2;
                            if ((f0 * f0 + f1 * f1) + f2 * f2 < dist + 1.0e-6) need_add = false;
                        }
                    }
                    c = need_add ? 1 : 0;
                }
            }
            code[i] = c;
        }
    }
    if (lx != nullptr) {  // (uniform)
        __shared__ unsigned wsum[256 / 64][2];
        unsigned v[2] = {c == 1 ? 1u : 0u, c == 2 ? 1u : 0u}, tot[2];
        block_excl_scan<2>(v, tot, wsum);
        if (i < n) lx[i] = make_uint2(v[0], v[1]);
        if (threadIdx.x == 0 && blockIdx.x * 256 < n) bt[blockIdx.x] = make_uint2(tot[0], tot[1]);
        if (threadIdx.x == 0 && blockIdx.x == 0) { *st_status = 0u; *st_apply = 0u; }  // kUpdOk; the batch's verdict is open
    }
}

}  // namespace fls
