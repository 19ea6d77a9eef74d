// kernels_exactsort.hpp -- libstdc++'s std::sort (introsort) of {key, value} records, compared by key only, reproduced on the
// device PERMUTATION FOR PERMUTATION (round 4).
//
// Why: pcl::VoxelGrid sums the points of a leaf in the order std::sort leaves records of equal leaf index in
// (include/common/pointcloud_utility.h:216-271 -> PCL 1.10 voxel_grid.hpp: std::sort of (idx, cloud_point_index) by idx).  That order
// is a property of introsort's partition sequence, not of the data, so a stable radix sort gives centroids that differ in the last
// float bits (kernels_voxelgrid.hpp CONTRACT) -- which is why the device filter was opt-in for three rounds.  With this sort in front
// of the same centroid kernel the device filter is bit-identical to the reference's and can be the default.
//
// Introsort = { median-of-three to first; unguarded Hoare partition around *first; recurse on [cut, last), loop on [first, cut) } until a
// range holds <= 16 records, then one insertion sort over everything (which never moves a record across a partition cut: left <= pivot
// <= right).  The partition is sequential as written, but its RESULT has a closed form (derivation + a CPU model checked against
// std::sort on 400 arrays: DESIGN.md 4, tests/host/exact_sort_model_test.cpp):
//   L-stops  L_1 < L_2 < ...   positions in [first+1, last) whose key >= pivot   (where the upward scan stops)
//   R-stops  R_1 > R_2 > ...   positions in [first+1, last) whose key <= pivot, then `first` itself   (downward scan)
//   the k-th swap exchanges L_k and R_k; swaps happen for k = 1 .. K*, K* = #{k : L_k < R_k} (monotone in k); every swap touches
//   positions no other swap touches;   cut = min(L_{K*+1}, R_{K*})  (terms that do not exist count as +inf).
// So one level = flag two predicates, rank them (prefix sums), scatter the two stop lists, swap K* disjoint pairs.  All ranges of one
// recursion depth are independent and processed together (level-synchronous); values ride along with their keys.
//   regime 1  ranges longer than a threshold: three launches per level over all such ranges (es_level_begin: cuts + children of the previous
//             level, medians of this one; es_count_scatter: stop lists through per-tile counts, a tile looks back at its predecessors'
//             published counts -- es_count / es_scatter are the two-launch form it replaced; es_swap).  Round 5: for clouds up to
//             kEsTaskMax records the levels above 32,768 records (FLS_ES_BIG) are PRE-ENQUEUED without a host round trip (device_voxelgrid.hpp
//             fused_launch: a level of launches costs ~16 us whatever the size, one workgroup 3 us + 0.38 us per thousand records); beyond
//             kEsTaskMax the host steers the levels through a mailbox as in round 4;
//   regime 2  shorter ranges are TASKS of the persistent es_task_kernel: a workgroup partitions a range longer than lds_cap (2,048 / 8,192 by cloud size) records out
//             of global memory and hands one child to the queue, takes a range of <= lds_cap records into LDS, runs ALL its remaining levels
//             there (sub-ranges above kEsCoop by the whole workgroup, the rest as wave tasks through a ticket queue: es_phase_b), then
//             ranks the records of every final <= 16 block (= the insertion sort) and writes them back.
// The heap-sort fallback of introsort (recursion deeper than 2 log2 n) IS reproduced for ranges that fit into LDS (es_heap_sort: real scans
// reach it routinely); only a range still longer than lds_cap records at depth 0 makes the sort report failure (the caller takes the host
// path).
#pragma once
#include "device_common.hpp"

namespace fls {

// (A/B r05, fls_match from host buffers: 8192 / 4096 / 2048 / 1024 records -> NDT 0.533 / 0.505 / 0.491 / 0.619 ms, ICP 0.443 / 0.415 / 0.408 / 0.404 ms.
// One CU sorts a range at ~13 us per thousand records -- 16 waves, ~1.1 us of dependent LDS round trips per partition -- while splitting a range
// in two out of global memory costs 3 us + 0.38 us per thousand: smaller LDS ranges on more CUs win until the queue traffic takes over)
// Round 6: the range a workgroup takes into LDS is a RUN-TIME parameter of es_task_kernel (`lds_cap`), chosen by the size of the cloud
// (DeviceExactSort::lds_cap_for): 2,048 records for clouds up to kEsTaskMax (every source scan: more CUs busy), 4,096 beyond (the map-side filters
// of the kd-tree kinds, 0.2-1.6 M records; measured 2,048 / 4,096 / 8,192: planar deque 0.83 / 0.80 / 0.80 ms, IcpOptimized deque 0.48 / 0.48 / 0.52 ms,
// profiles/r06_g_*).  FLS_ES_LDS = the capacity the LDS arrays are sized for.
// (The regression between BENCH_r04 and BENCH_r05 -- LoamFull keyframe update 2.25 -> 5.54 ms -- showed up as "2,048-record ranges are 3.6x slower than
// 8,192-record ranges on 1.55 M records"; the cause was not the range size but the task queue's pop, see es_task_kernel: the ticket queue.)
#ifndef FLS_ES_LDS
#define FLS_ES_LDS 4096
#endif
constexpr int kEsLds = FLS_ES_LDS;  // records the LDS arrays hold (the largest lds_cap)
constexpr int kEsLdsSmall = 2048;   // default lds_cap for clouds up to kEsTaskMax records
constexpr int kEsTaskMax = 131072;  // ranges up to here are tasks of the persistent kernel; longer ones go through the level-synchronous launches
constexpr int kEsThreshold = 16;    // _S_threshold
constexpr int kEsTile = 2048, kEsBlock = 256, kEsItems = kEsTile / kEsBlock;
constexpr int kEsMaxSeg = 192;      // regime-1 ranges of one level (n / hand-over threshold: 128 for 4 Mi records at 32,768, with room)

struct EsSeg { unsigned first, last; int depth; unsigned pivot, nL, nR, K, tile0; };
struct EsWork { unsigned first, last; int depth, pad; };
struct alignas(16) EsQueue { unsigned head, tail, open, n_init; };  // the task kernel's queue (entries below n_init are ready without a flag)
struct EsState {
    unsigned n_cur;     // regime-1 ranges of the level in flight
    unsigned n_tiles;   // their tiles
    unsigned n_work;    // ranges handed to regime 2 so far
    unsigned fail;      // 1: introsort would heap-sort / a table overflowed -> the caller takes the host path
    unsigned level;     // regime-1 levels run
    unsigned pad[3];
};
// what the host polls (host-mapped): written by every es_level_begin
struct EsMailbox { unsigned seq, n_cur, n_work, fail; unsigned mark[12]; unsigned lvl[32][8]; unsigned wg[256][8]; };  // wg[b] (FLS_ES_DEBUG): workgroup b's 100 MHz ticks waiting for a task | in partitions out of global memory | in LDS ranges, its task count, its longest LDS range (ticks, records), its longest chain of partitions out of global memory (ticks, first range's records)  // mark / lvl: stage stamps of es_task_kernel (diagnostics, FLS_ES_DEBUG): lvl[i] = {range size, 100 MHz stamps of the phases of workgroup 0's i-th partition out of global memory}

// (Round 6, built and REMOVED: "write-through records" -- every key[] / val[] access of the task kernel as an agent-scope relaxed atomic (sc1), the
// hand-over between workgroups without release / acquire fences.  Times equal to the fenced form within 2 % (profiles/r06_vg_large_cloud_filters.txt), and
// WRONG under concurrency: with three or four batch lanes sorting at once, ~15 % of the IcpOptimized jobs got a source cloud with too many leaves -- an
// unsorted piece (tools/dbg_batch_stress.py: 44 of 80 repetitions with a mismatch, 0 of 80 fenced).  The fences stay.)
__device__ __forceinline__ void es_swap_rec(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned a, const unsigned b) {
    const unsigned ka = key[a], kb = key[b], va = val[a], vb = val[b];
    key[a] = kb; key[b] = ka; val[a] = vb; val[b] = va;
}
// std::__move_median_to_first(result = first, a = first + 1, b = mid, c = last - 1) on the keys; returns the pivot key.
// All four records are loaded before anything is decided (ONE memory round trip instead of three dependent ones -- candidates, then the
// two records to swap, then the pivot read back: es_level_begin is a chain of such hops, 5.3 us per level in the trace).
__device__ __forceinline__ unsigned es_median_to_first(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned first, const unsigned last) {
    const unsigned a = first + 1u, b = first + (last - first) / 2u, c = last - 1u;
    const unsigned ka = key[a], kb = key[b], kc = key[c], kf = key[first];
    const unsigned va = val[a], vb = val[b], vc = val[c], vf = val[first];
    unsigned med, km, vm;
    if (ka < kb) { if (kb < kc) { med = b; km = kb; vm = vb; } else if (ka < kc) { med = c; km = kc; vm = vc; } else { med = a; km = ka; vm = va; } }
    else if (ka < kc) { med = a; km = ka; vm = va; }
    else if (kb < kc) { med = c; km = kc; vm = vc; }
    else { med = b; km = kb; vm = vb; }
    key[first] = km; val[first] = vm; key[med] = kf; val[med] = vf;  // (a, b, c are distinct positions of a range of more than 16 records)
    return km;
}

// ---- regime 1 ----------------------------------------------------------------------------------------------------------------------
// One workgroup: finish the previous level (cut of every range from its K*, children -> next level or the LDS work list), then open
// the new level (median of three, pivot, tile map).  `prev` / `cur` are the two range tables, swapped by the host every level.
__global__ void __launch_bounds__(256)
es_level_begin(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned n, const EsSeg* __restrict__ prev, EsSeg* __restrict__ cur,
               EsWork* __restrict__ work, const unsigned work_cap, const unsigned* __restrict__ Lp, const unsigned* __restrict__ Rl,
               EsState* __restrict__ st, EsMailbox* __restrict__ mb, const unsigned seq, const int first_level, unsigned* __restrict__ tile_seg,
               const unsigned tile_cap,
               // round 5 (the pre-enqueued top levels of the one-stream VoxelGrid, DeviceExactSort::fused_launch): ranges longer than `big` stay
               // level-synchronous; final_level: every child becomes a task and the task kernel's queue is written here; *skip != 0: nothing to sort
               const unsigned big, const int final_level, EsQueue* __restrict__ q_out, const unsigned* __restrict__ skip) {
    __shared__ unsigned s_ncur, s_nwork, s_fail, s_tiles[kEsMaxSeg], s_total;
    __shared__ EsSeg s_cur[kEsMaxSeg];  // the new level's ranges: completed here (pivot, tile map), written out once
    if (skip != nullptr && *skip != 0u) {
        if (threadIdx.x == 0) {
            st->n_cur = 0u; st->n_tiles = 0u; st->n_work = 0u;
            if (q_out != nullptr) *q_out = EsQueue{0u, 0u, 0u, 0u};
        }
        return;
    }
    if (threadIdx.x == 0) { s_ncur = 0u; s_nwork = first_level ? 0u : st->n_work; s_fail = first_level ? 0u : st->fail; }
    __syncthreads();
    const unsigned stay = final_level ? 0xFFFFFFFFu : big;
    auto child = [&](const unsigned f, const unsigned l, const int depth) {
        const unsigned m = l - f;
        if (m < 2u) return;
        if (m > stay) {
            const unsigned slot = atomicAdd(&s_ncur, 1u);
            if (slot < (unsigned)kEsMaxSeg) s_cur[slot] = EsSeg{f, l, depth, 0u, 0u, 0u, 0u, 0u};
            else s_fail = 1u;
        } else {
            const unsigned slot = atomicAdd(&s_nwork, 1u);
            if (slot < work_cap) work[slot] = EsWork{f, l, depth, 0};
            else s_fail = 1u;
        }
    };
    if (first_level) {
        if (threadIdx.x == 0) {
            int lg = 0;
            while ((2u << lg) <= n) ++lg;  // floor(log2 n)
            child(0u, n, 2 * lg);
        }
    } else {
        const unsigned np = st->n_cur;
        for (unsigned s = threadIdx.x; s < np; s += 256u) {
            const EsSeg g = prev[s];
            const unsigned INF = 0xFFFFFFFFu;
            const unsigned a = g.K < g.nL ? Lp[g.first + g.K] : INF;
            const unsigned b = g.K >= 1u ? (g.K - 1u < g.nR ? Rl[g.first + g.nR - g.K] : g.first) : INF;  // R_{K*} (1-based) = Rl[first + nR - K*]
            const unsigned cut = a < b ? a : b;
            child(cut, g.last, g.depth - 1);
            child(g.first, cut, g.depth - 1);
        }
    }
    __syncthreads();
    const unsigned nc = s_ncur < (unsigned)kEsMaxSeg ? s_ncur : (unsigned)kEsMaxSeg;
    for (unsigned s = threadIdx.x; s < nc; s += 256u) {
        EsSeg g = s_cur[s];
        if (g.depth == 0) { s_fail = 1u; g.pivot = key[g.first]; }  // (introsort switches to heap sort here)
        else g.pivot = es_median_to_first(key, val, g.first, g.last);
        s_cur[s] = g;
        s_tiles[s] = (g.last - g.first - 1u + (unsigned)kEsTile - 1u) / (unsigned)kEsTile;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix of the tile counts (LDS, <= kEsMaxSeg entries)
        unsigned t = 0u;
        for (unsigned s = 0; s < nc; ++s) { const unsigned c = s_tiles[s]; s_tiles[s] = t; t += c; }
        s_total = t;
        if (t > tile_cap) s_fail = 1u;
    }
    __syncthreads();
    if (!s_fail) {
        for (unsigned s = threadIdx.x; s < nc; s += 256u) {
            EsSeg g = s_cur[s];
            g.tile0 = s_tiles[s];
            cur[s] = g;
        }
        // tile -> range map (one load per workgroup in the launches that follow), written by ALL threads: a keyframe deque's first levels are one or two
        // ranges of 760 tiles, which one thread per range wrote one by one (round 6: es_level_begin 7-8 us per level on 1.55 M records)
        const unsigned total = s_total;
        for (unsigned q = threadIdx.x; q < total; q += 256u) {
            unsigned lo = 0u, hi = nc;  // the last s with s_tiles[s] <= q
            while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (s_tiles[mid] <= q) lo = mid; else hi = mid; }
            tile_seg[q] = lo;
        }
    }
    if (threadIdx.x == 0) {
        const unsigned t = s_total;
        st->n_cur = s_fail ? 0u : nc;
        st->n_tiles = s_fail ? 0u : t;
        st->n_work = s_nwork < work_cap ? s_nwork : work_cap;
        st->fail = s_fail;
        st->level = first_level ? 0u : st->level + 1u;
        if (q_out != nullptr && final_level) { const unsigned nw = st->n_work; *q_out = EsQueue{0u, nw, nw, nw}; }
        if (mb != nullptr) {  // (the host steers the levels only beyond kEsTaskMax records; the pre-enqueued levels of the one-stream form pass nullptr)
            __hip_atomic_store(&mb->n_cur, st->n_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->n_work, st->n_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->fail, s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// workgroup -> (range, tile)
__device__ __forceinline__ bool es_locate(const EsSeg* __restrict__ cur, const EsState* __restrict__ st, const unsigned* __restrict__ tile_seg, unsigned& seg,
                                          unsigned& tile) {
    if (blockIdx.x >= st->n_tiles) return false;
    seg = tile_seg[blockIdx.x];
    tile = blockIdx.x - cur[seg].tile0;
    return true;
}
// per-tile counts of the two predicates
__global__ void __launch_bounds__(kEsBlock)
es_count_kernel(const unsigned* __restrict__ key, const EsSeg* __restrict__ cur, const EsState* __restrict__ st, const unsigned* __restrict__ tile_seg,
                uint2* __restrict__ tile_cnt) {
    __shared__ unsigned wsum[kEsBlock / 64][2];
    unsigned seg, tile;
    if (!es_locate(cur, st, tile_seg, seg, tile)) return;
    const EsSeg g = cur[seg];
    const unsigned base = g.first + 1u + tile * (unsigned)kEsTile;
    unsigned cl = 0u, cr = 0u;
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned i = base + q * (unsigned)kEsBlock + threadIdx.x;
        if (i < g.last) { const unsigned k = key[i]; cl += k >= g.pivot ? 1u : 0u; cr += k <= g.pivot ? 1u : 0u; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cl += __shfl_xor(cl, o, 64); cr += __shfl_xor(cr, o, 64); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = cl; wsum[threadIdx.x >> 6][1] = cr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0u, b = 0u;
        for (int w = 0; w < kEsBlock / 64; ++w) { a += wsum[w][0]; b += wsum[w][1]; }
        tile_cnt[blockIdx.x] = make_uint2(a, b);
    }
}
// stop lists: L-stops ascending at Lp[first + rank], R-stops ascending at Rl[first + rank] (the k-th from the right is Rl[first + nR - 1 - k])
__global__ void __launch_bounds__(kEsBlock)
es_scatter_kernel(const unsigned* __restrict__ key, EsSeg* __restrict__ cur, const EsState* __restrict__ st, const unsigned* __restrict__ tile_seg,
                  const uint2* __restrict__ tile_cnt, unsigned* __restrict__ Lp, unsigned* __restrict__ Rl) {
    __shared__ unsigned wsum4[kEsBlock / 64][4], wsum2[kEsBlock / 64][2];
    unsigned seg, tile;
    if (!es_locate(cur, st, tile_seg, seg, tile)) return;
    const EsSeg g = cur[seg];
    const unsigned ntile = (g.last - g.first - 1u + (unsigned)kEsTile - 1u) / (unsigned)kEsTile;
    unsigned pre[2], all[2];
    block_prefix_total<2>(tile_cnt + g.tile0, (int)ntile, (int)tile, pre, all, wsum4);
    if (tile == 0u && threadIdx.x == 0) { cur[seg].nL = all[0]; cur[seg].nR = all[1]; cur[seg].K = 0u; }
    // a thread owns kEsItems CONSECUTIVE positions (ranks follow positions)
    const unsigned base = g.first + 1u + tile * (unsigned)kEsTile + threadIdx.x * (unsigned)kEsItems;
    unsigned kk[kEsItems];
    unsigned v[2] = {0u, 0u}, tot[2];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned i = base + q;
        kk[q] = i < g.last ? key[i] : 0u;
        if (i < g.last) { v[0] += kk[q] >= g.pivot ? 1u : 0u; v[1] += kk[q] <= g.pivot ? 1u : 0u; }
    }
    block_excl_scan<2>(v, tot, wsum2);
    unsigned rl = g.first + pre[0] + v[0], rr = g.first + pre[1] + v[1];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned i = base + q;
        if (i < g.last) {
            if (kk[q] >= g.pivot) Lp[rl++] = i;
            if (kk[q] <= g.pivot) Rl[rr++] = i;
        }
    }
}
// es_count_kernel + es_scatter_kernel in ONE launch (round 5): a tile counts its two predicates, publishes the pair write-through together
// with the launch's epoch, and looks BACK at the tiles of its range in front of it -- one predecessor per thread, polled with loads that
// are served from memory -- instead of waiting for a launch boundary (a dependent launch costs ~4.7 us here, the look-back ~1 us).
// A tile only ever waits for LOWER workgroup ids of the same launch, and the hardware starts workgroups in id order: no tile waits for
// one that cannot run.  pub[t] = epoch << 32 | n_R-stops << 16 | n_L-stops (a tile holds 2,048 records); the epoch never repeats, so a
// word of an earlier level / sort is never taken for this launch's.  "The hardware starts workgroups in id order" is what dispatch does, not a guarantee
// (ADVICE r5): with other streams' kernels competing for the CUs a predecessor could in principle still be waiting for a slot while its successor polls.
// An atomic ticket per workgroup would remove the assumption at ~1 us per launch on the critical path of every level; instead the poll gives up after
// ~50 ms (40 k polls) and fails the sort, and the caller takes the exact host filter: correct either way, never a stall of seconds.
__global__ void __launch_bounds__(kEsBlock)
es_count_scatter_kernel(const unsigned* __restrict__ key, EsSeg* __restrict__ cur, EsState* __restrict__ st, const unsigned* __restrict__ tile_seg,
                        unsigned long long* __restrict__ pub, const unsigned epoch, unsigned* __restrict__ Lp, unsigned* __restrict__ Rl) {
    __shared__ unsigned wsum2[kEsBlock / 64][2], wpre[kEsBlock / 64][3];
    unsigned seg, tile;
    if (!es_locate(cur, st, tile_seg, seg, tile)) return;
    const EsSeg g = cur[seg];
    const unsigned ntile = (g.last - g.first - 1u + (unsigned)kEsTile - 1u) / (unsigned)kEsTile;
    // a thread owns kEsItems CONSECUTIVE positions (ranks follow positions)
    const unsigned base = g.first + 1u + tile * (unsigned)kEsTile + threadIdx.x * (unsigned)kEsItems;
    unsigned kk[kEsItems];
    unsigned v[2] = {0u, 0u}, tot[2];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned i = base + q;
        kk[q] = i < g.last ? key[i] : 0u;
        if (i < g.last) { v[0] += kk[q] >= g.pivot ? 1u : 0u; v[1] += kk[q] <= g.pivot ? 1u : 0u; }
    }
    block_excl_scan<2>(v, tot, wsum2);
    if (threadIdx.x == 0)
        __hip_atomic_store(&pub[blockIdx.x], ((unsigned long long)epoch << 32) | ((unsigned long long)tot[1] << 16) | (unsigned long long)tot[0],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned p0 = 0u, p1 = 0u, dry = 0u;
    for (unsigned t = threadIdx.x; t < tile; t += (unsigned)kEsBlock) {
        unsigned long long w = 0ull;
        unsigned guard = 0u;
        for (;;) {
            w = __hip_atomic_load(&pub[g.tile0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == epoch) break;
            if (++guard > 40000u) { dry = 1u; break; }  // (~40-80 ms; 4 M polls = seconds until round 6)
            __builtin_amdgcn_s_sleep(1);
        }
        p0 += (unsigned)w & 0xFFFFu; p1 += (unsigned)(w >> 16) & 0xFFFFu;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o, 64); p1 += __shfl_xor(p1, o, 64); dry |= __shfl_xor(dry, o, 64); }
    if ((threadIdx.x & 63) == 0) { wpre[threadIdx.x >> 6][0] = p0; wpre[threadIdx.x >> 6][1] = p1; wpre[threadIdx.x >> 6][2] = dry; }
    __syncthreads();
    unsigned pre0 = 0u, pre1 = 0u, bad = 0u;
#pragma unroll
    for (int w = 0; w < kEsBlock / 64; ++w) { pre0 += wpre[w][0]; pre1 += wpre[w][1]; bad |= wpre[w][2]; }
    if (bad) {
        if (threadIdx.x == 0) __hip_atomic_store(&st->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tile + 1u == ntile && threadIdx.x == 0) { cur[seg].nL = pre0 + tot[0]; cur[seg].nR = pre1 + tot[1]; cur[seg].K = 0u; }
    unsigned rl = g.first + pre0 + v[0], rr = g.first + pre1 + v[1];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned i = base + q;
        if (i < g.last) {
            if (kk[q] >= g.pivot) Lp[rl++] = i;
            if (kk[q] <= g.pivot) Rl[rr++] = i;
        }
    }
}
// the K* swaps of every range (pair k = L_k <-> R_k, disjoint), and K* itself (the one k with cond(k) && !cond(k + 1))
__global__ void __launch_bounds__(kEsBlock)
es_swap_kernel(unsigned* __restrict__ key, unsigned* __restrict__ val, EsSeg* __restrict__ cur, const EsState* __restrict__ st,
               const unsigned* __restrict__ tile_seg, const unsigned* __restrict__ Lp, const unsigned* __restrict__ Rl) {
    unsigned seg, tile;
    if (!es_locate(cur, st, tile_seg, seg, tile)) return;
    const EsSeg g = cur[seg];
    // (round 5: three phases with every item's loads in flight together -- stop positions, then records, then stores.  The item-by-item loop it
    // replaces chained eight dependent three-hop round trips per thread: 9-11 us per launch in profiles/r05_e_ndt_top_levels_sequence.txt)
    const unsigned span = g.last - g.first - 1u;
    const unsigned kmax = g.nL < g.nR + 1u ? g.nL : g.nR + 1u;  // pairs beyond min(nL, nR + 1) cannot hold
    unsigned a[kEsItems + 0], b[kEsItems + 0];
    bool c[kEsItems], cn[kEsItems];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned k = tile * (unsigned)kEsTile + q * (unsigned)kEsBlock + threadIdx.x;
        const bool in = k < span && k < kmax;
        a[q] = in ? Lp[g.first + k] : 0u;
        b[q] = in ? (k < g.nR ? Rl[g.first + g.nR - 1u - k] : g.first) : 0u;
    }
    unsigned an[kEsItems], bn[kEsItems];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {  // the pair after it (the K* test): the same words one thread further, L1-resident
        const unsigned k = tile * (unsigned)kEsTile + q * (unsigned)kEsBlock + threadIdx.x + 1u;
        const bool in = k < span && k < kmax;
        an[q] = in ? Lp[g.first + k] : 0u;
        bn[q] = in ? (k < g.nR ? Rl[g.first + g.nR - 1u - k] : g.first) : 0u;
    }
    unsigned ka[kEsItems], kb[kEsItems], va[kEsItems], vb[kEsItems];
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned k = tile * (unsigned)kEsTile + q * (unsigned)kEsBlock + threadIdx.x;
        c[q] = k < span && k < kmax && a[q] < b[q];
        cn[q] = k + 1u < span && k + 1u < kmax && an[q] < bn[q];
        if (c[q]) { ka[q] = key[a[q]]; kb[q] = key[b[q]]; va[q] = val[a[q]]; vb[q] = val[b[q]]; }
    }
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        if (!c[q]) continue;
        const unsigned k = tile * (unsigned)kEsTile + q * (unsigned)kEsBlock + threadIdx.x;
        key[a[q]] = kb[q]; key[b[q]] = ka[q]; val[a[q]] = vb[q]; val[b[q]] = va[q];
        if (!cn[q]) cur[seg].K = k + 1u;
    }
}

// ---- regimes 2 + 3: ONE persistent launch, a task queue of ranges -------------------------------------------------------------------
// A level-synchronous sweep pays for the DEEPEST branch at every level, and introsort on LiDAR leaf indices is deep and lopsided
// (median-of-three on piecewise-monotone keys: 16 levels before the last range of a 115 k-point scan fell below 4,096 records, 23 more
// inside it).  From kEsTaskMax records down, ranges are therefore TASKS: a workgroup pops a range from a global queue and
//   * partitions it itself out of global memory (16 waves, each over a contiguous slice in coalesced rounds of 64 with ballot ranks:
//     counts -> wave bases -> stop lists -> K* disjoint swaps) and pushes the two children, or
//   * (range <= lds_cap) takes it into LDS, where its WAVES run the same partition on sub-ranges from a local queue -- no workgroup
//     barrier per level, every branch advances at its own pace -- then ranks the records of every final <= 16 block (= the insertion
//     sort) and writes the range back.
// Hand-off between workgroups (possibly on different XCDs): the producer writes back its L2 (agent-scope release) before it publishes a
// child, the consumer invalidates (agent-scope acquire) after it pops one.  Termination: a counter of open tasks.
#ifndef FLS_ES_UNROLL
#define FLS_ES_UNROLL 8  // independent key loads a lane keeps in flight in the passes of a partition out of global memory (A/B at the end of round 4,
                         // NDT fls_match from host buffers: 8 -> 0.682 ms, 16 -> 0.676, 24 -> 0.691, 40 -> 0.934 (spills): not what bounds the stage)
#endif
constexpr int kEsTaskThreads = 1024, kEsTaskWaves = kEsTaskThreads / 64;
#ifndef FLS_ES_COOP
#define FLS_ES_COOP 4096  // (A/B r05: 1024 / 2048 / 4096 / 8192 -> ICP call 0.466 / 0.449 / 0.444 / 0.478 ms)
#endif
#ifndef FLS_ES_SHARE
#define FLS_ES_SHARE 64    // (A/B r05: 384 / 192 / 96 / 64 / 48 / 32 -> 0.497 / 0.451 / 0.444 / 0.445 / 0.445 / 0.475 ms)
#endif
constexpr int kEsCoop = FLS_ES_COOP < FLS_ES_LDS ? FLS_ES_COOP : FLS_ES_LDS;    // sub-ranges of an LDS range longer than this are partitioned by the whole workgroup, shorter ones by single waves (never more than lds_cap: such ranges do not exist)
constexpr int kEsShare = FLS_ES_SHARE;  // a wave hands children longer than this to the workgroup's queue (another wave takes them), shorter ones stay on its own stack
constexpr int kEsStack = 64;     // pending workgroup-level sub-ranges (disjoint, each > kEsCoop records: at most kEsLds / kEsCoop)
constexpr int kEsWaveStack = 48; // a wave's depth-first stack (smaller child first: <= log2(kEsCoop) + 1 pending ranges)
constexpr int kEsLocalQ = 1024;  // sub-range tasks of one LDS range (<= 2 per partition, <= kEsLds / 17 partitions)

__device__ __forceinline__ unsigned es_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// push a range (>= 2 records) onto the global queue; the caller has released its writes
__device__ __forceinline__ void es_push(EsQueue* __restrict__ q, EsWork* __restrict__ tasks, unsigned* __restrict__ ready, const unsigned cap, EsState* __restrict__ st,
                                        const unsigned f, const unsigned l, const int depth) {
    if (l - f < 2u) return;
    __hip_atomic_fetch_add(&q->open, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (no return value: does not wait)
    const unsigned slot = atomicAdd(&q->tail, 1u);
    if (slot >= cap) { atomicExch(&st->fail, 1u); atomicSub(&q->open, 1u); return; }
    tasks[slot] = EsWork{f, l, depth, 0};  // (plain stores: the release below orders them, and the sorted records, before the flag)
    __hip_atomic_store(&ready[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// std::__partial_sort(first, last, last) = __make_heap + __sort_heap of libstdc++ (bits/stl_heap.h), what introsort falls back to when a
// range exhausts the depth limit 2 log2 n.  NOT a corner case here: ring-major LiDAR leaf indices are piecewise monotone, median-of-three
// splits them lopsidedly, and most 115 k-point scans end with a few ranges of 30-70 records at depth 0 (tools/es_decline_probe.py, the CPU
// model) -- the reference's own std::sort heap-sorts them, so the device does too: ONE lane runs the restated routines on the range in LDS
// (the order heap sort leaves equal keys in is as implementation-defined as introsort's; tests/host/exact_sort_model_test.cpp checks the
// restatement against std::sort on 1,500 such ranges).  Keys compare with '<' only, values ride along.
// `St` = how a record is stored: es_store_plain (one lane runs the routine) or es_store_lane0 (every lane of a wave runs it in lock-step on the
// same words -- wave-uniform control flow -- and only lane 0's stores land on the data, es_phase_b)
struct es_store_plain { __device__ __forceinline__ void operator()(unsigned* p, const unsigned x) const { *p = x; } };
template <class St>
__device__ __forceinline__ void es_heap_adjust(unsigned* __restrict__ k, unsigned* __restrict__ v, int hole, const int len, const unsigned vk, const unsigned vv, const St st) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (k[second] < k[second - 1]) second--;
        const unsigned a = k[second], b = v[second];
        st(&k[hole], a); st(&v[hole], b);
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        const unsigned a = k[second - 1], b = v[second - 1];
        st(&k[hole], a); st(&v[hole], b);
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;  // __push_heap
    while (hole > top && k[parent] < vk) { const unsigned a = k[parent], b = v[parent]; st(&k[hole], a); st(&v[hole], b); hole = parent; parent = (hole - 1) / 2; }
    st(&k[hole], vk); st(&v[hole], vv);
}
template <class St>
__device__ __forceinline__ void es_heap_sort(unsigned* __restrict__ k, unsigned* __restrict__ v, const int len, const St st) {  // k, v: the range's first record
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) { es_heap_adjust(k, v, parent, len, k[parent], v[parent], st); if (parent == 0) break; }
    for (int last = len - 1; last >= 1; --last) {
        const unsigned vk = k[last], vv = v[last];
        const unsigned a = k[0], b = v[0];
        st(&k[last], a); st(&v[last], b);
        es_heap_adjust(k, v, 0, last, vk, vv, st);
    }
}
__device__ __forceinline__ void es_heap_sort(unsigned* __restrict__ k, unsigned* __restrict__ v, const int len) { es_heap_sort(k, v, len, es_store_plain{}); }

// LDS of es_task_kernel at namespace scope: the wave-task phase is a function of its own (es_phase_b) that names these arrays directly
namespace es_lds {
__shared__ unsigned sk[kEsLds], sv[kEsLds];
__shared__ unsigned short lp[kEsLds], rl[kEsLds];
__shared__ unsigned bmask[kEsLds / 32 + 2];  // bit i: a partition cut (or the range's ends) lies in front of record i
__shared__ unsigned ws_t[kEsTaskWaves][kEsWaveStack];  // per-wave depth-first stacks of phase B
__shared__ signed char ws_d[kEsTaskWaves][kEsWaveStack];
__shared__ unsigned qt[kEsLocalQ];  // local task = first | last << 16, published with one release store (0 = not yet)
__shared__ signed char qd[kEsLocalQ];
__shared__ unsigned lq_head, lq_tail, lq_open, s_sp;
__shared__ unsigned short sa_f[kEsStack], sa_l[kEsStack];
__shared__ signed char sa_d[kEsStack];
__shared__ unsigned s_wl[kEsTaskWaves], s_wr[kEsTaskWaves], s_cnt[kEsTaskWaves];
__shared__ unsigned s_first, s_last, s_pivot, s_state, s_fail;
__shared__ int s_depth;
// per-lane scratch words: what lanes 1..63 write where only lane 0's store counts (es_phase_b: wave-uniform control flow WITHOUT sixty-four
// stores to one address, which the LDS would serialise)
__shared__ unsigned dmy_u[64];
__shared__ unsigned short dmy_h[64];
__shared__ signed char dmy_c[64];
}  // namespace es_lds
struct es_store_lane0 {
    int lane;
    __device__ __forceinline__ void operator()(unsigned* p, const unsigned x) const { *(lane == 0 ? p : &es_lds::dmy_u[lane]) = x; }
};

// phase B of an LDS range (see es_task_kernel): the remaining sub-ranges (<= kEsCoop records) are tasks of a TICKET QUEUE in LDS (round 5).
// A wave takes a ticket, waits until that slot is published (or until no task is open any more: nothing will ever be published), partitions
// the range and keeps the SMALLER child on its private stack; a larger child longer than kEsShare records goes back to the queue, where an idle
// wave picks it up.  (Round 4 gave every queued range with its whole subtree to one wave: <= 8 tasks for 16 waves, 40-80 us on the slowest
// wave -- profiles/r05_a_vg_*_before_exact_sort_stamps.log.)  lq_open = queue tasks not yet finished, subtrees included.
//
// CONTROL FLOW IS WAVE-UNIFORM ON PURPOSE.  A divergent `if (lane == 0) { atomic ... }` region that ends at a loop's back edge was compiled
// (ROCm 7.2, gfx950) into a loop whose lane 0 and lanes 1..63 take the back edge SEPARATELY; the convergent operation at the loop's top
// (readfirstlane / a __shfl of the ticket) then ran once per group, lanes 1..63 re-read ticket 0 and processed that task again -- partitions
// with a partial exec mask (nL == 0 on a 76-record range) or an endless loop.  tools/experiments/lds_ticket_queue_test.hip reproduces it in
// forty lines.  So: every lane executes every atomic (only lane 0 adds a non-zero value), every lane stores the wave's scalars (same value,
// same address), scalars read back from LDS go through readfirstlane, and the median / heap-sort steps are executed by all lanes in
// lock-step on the same words (reads of an instruction precede its writes: identical results).  A function of its own, not inlined: the
// task kernel around it is 5,000 instructions deep in live scalars.
__device__ __attribute__((noinline)) void es_phase_b(const unsigned m) {
    using namespace es_lds;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // lane 0's address, a private scratch word for every other lane (one LDS pass instead of sixty-four serialised stores / atomics)
    auto u0 = [&](unsigned* q) { return lane == 0 ? q : &dmy_u[lane]; };
    auto c0 = [&](signed char* q) { return lane == 0 ? q : &dmy_c[lane]; };
    const es_store_lane0 st0{lane};
    auto rfl = [](const unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (;;) {
        const unsigned ticket = rfl(atomicAdd(u0(&lq_head), 1u));
        unsigned word = 0u;
        if (ticket < (unsigned)kEsLocalQ) {
            for (unsigned spin = 0;; ++spin) {
                word = rfl(__hip_atomic_load(&qt[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (word != 0u) break;
                if (rfl(__hip_atomic_load(&lq_open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) break;
                if (rfl(__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) break;
                if (spin > 400000u) { __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }  // watchdog (tens of ms): never hang
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (word == 0u) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        *u0(&ws_t[w][0]) = word;
        *c0(&ws_d[w][0]) = qd[ticket];
        int sp = 1;
        while (sp > 0) {
            --sp;
            wave_sync();
            const unsigned task_word = rfl(ws_t[w][sp]);
            const int d = (int)rfl((unsigned)(unsigned char)ws_d[w][sp]);  // (depths are 0 .. 127)
            const unsigned f = task_word & 0xffffu, l = task_word >> 16;
            if (f >= l || l > m) { __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }  // (never: a corrupted word must not become wild LDS indices)
            if (d == 0) {  // depth limit: heap sort (every lane runs it in lock-step on the same words); the range is sorted, every record a final block of its own
                es_heap_sort(sk + f, sv + f, (int)(l - f), st0);
                wave_sync();
                for (unsigned i = f + 1u + (unsigned)lane; i < l; i += 64u) atomicOr(&bmask[i >> 5], 1u << (i & 31u));
                continue;
            }
            unsigned p;
            {   // std::__move_median_to_first(first, first + 1, mid, last - 1): loads first, then the four stores
                const unsigned a = f + 1u, b = f + (l - f) / 2u, c = l - 1u;
                const unsigned ka = sk[a], kb = sk[b], kc = sk[c];
                unsigned med;
                if (ka < kb) { if (kb < kc) med = b; else if (ka < kc) med = c; else med = a; }
                else if (ka < kc) med = a;
                else if (kb < kc) med = c;
                else med = b;
                med = rfl(med);
                const unsigned k0 = sk[f], v0 = sv[f], km = sk[med], vm = sv[med];
                wave_sync();
                st0(&sk[f], km); st0(&sv[f], vm); st0(&sk[med], k0); st0(&sv[med], v0);
                p = rfl(km);
            }
            wave_sync();
            // one pass: both stop lists with left ranks (the running counts are the ranks); four rounds of reads in flight
            unsigned nL = 0u, nR = 0u;
            for (unsigned base = f + 1u; base < l; base += 64u * 4u) {
                unsigned kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const unsigned i = base + 64u * u + lane; kk[u] = i < l ? sk[i] : 0u; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned i = base + 64u * u + lane;
                    const bool fl = i < l && kk[u] >= p, fr = i < l && kk[u] <= p;
                    const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
                    if (fl) lp[f + nL + (unsigned)__popcll(ml & lt_mask)] = (unsigned short)i;
                    if (fr) rl[f + nR + (unsigned)__popcll(mr & lt_mask)] = (unsigned short)i;
                    nL += (unsigned)__popcll(ml); nR += (unsigned)__popcll(mr);
                }
            }
            wave_sync();
            unsigned K = 0u;
            for (unsigned k0 = 0u; k0 < nL; k0 += 64u) {
                const unsigned k = k0 + lane;
                bool c = false;
                unsigned a = 0u, b = 0u;
                if (k < nL) { a = lp[f + k]; b = k < nR ? (unsigned)rl[f + nR - 1u - k] : f; c = a < b; }
                if (c) { const unsigned ka = sk[a], va = sv[a]; sk[a] = sk[b]; sv[a] = sv[b]; sk[b] = ka; sv[b] = va; }
                const unsigned long long mc = __ballot(c);
                K += (unsigned)__popcll(mc);
                if (mc != ~0ull) break;  // (monotone: nothing beyond the first false)
            }
            wave_sync();
            // cut + children (scalars)
            const unsigned INF = 0xFFFFFFFFu;
            const unsigned ca = K < nL ? (unsigned)lp[f + K] : INF;
            const unsigned cb = K >= 1u ? (K - 1u < nR ? (unsigned)rl[f + nR - K] : f) : INF;
            const unsigned cut = rfl(ca < cb ? ca : cb);
            if (cut <= f || cut >= l) { __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }  // (never)
            atomicOr(u0(&bmask[cut >> 5]), 1u << (cut & 31u));
            // the SMALLER child next on this wave (the stack stays logarithmic); the larger one to the queue when it is worth another wave's while
            const unsigned m0 = cut - f, m1 = l - cut;
            const unsigned big_f = m0 >= m1 ? f : cut, big_l = m0 >= m1 ? cut : l, sm_f = m0 >= m1 ? cut : f, sm_l = m0 >= m1 ? l : cut;
            if (big_l - big_f > (unsigned)kEsThreshold) {
                bool shared = false;
                if (big_l - big_f > (unsigned)kEsShare) {
                    atomicAdd(u0(&lq_open), 1u);
                    const unsigned s2 = rfl(atomicAdd(u0(&lq_tail), 1u));
                    if (s2 < (unsigned)kEsLocalQ) {
                        *c0(&qd[s2]) = (signed char)(d - 1);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's records and the depth before the word
                        __hip_atomic_store(u0(&qt[s2]), big_f | (big_l << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        shared = true;
                    } else atomicSub(u0(&lq_open), 1u);  // (queue full: the range stays on this wave's stack)
                }
                if (!shared) {
                    if (sp < kEsWaveStack) { *u0(&ws_t[w][sp]) = big_f | (big_l << 16); *c0(&ws_d[w][sp]) = (signed char)(d - 1); ++sp; }
                    else __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (sm_l - sm_f > (unsigned)kEsThreshold) {
                if (sp < kEsWaveStack) { *u0(&ws_t[w][sp]) = sm_f | (sm_l << 16); *c0(&ws_d[w][sp]) = (signed char)(d - 1); ++sp; }
                else __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        atomicSub(u0(&lq_open), 1u);  // this queue task and its private subtree are done
    }
}

__global__ void __launch_bounds__(kEsTaskThreads)
es_task_kernel(unsigned* __restrict__ key, unsigned* __restrict__ val, EsWork* __restrict__ tasks, unsigned* __restrict__ ready, const unsigned cap,
               EsQueue* __restrict__ q, unsigned* __restrict__ Lp, unsigned* __restrict__ Rl, EsState* __restrict__ st, EsMailbox* __restrict__ dbg,
               const unsigned init_n /* != 0: the whole array [0, init_n) is workgroup 0's first task (no begin launch, no queue entry) */,
               const unsigned* __restrict__ skip /* nullable; *skip != 0: the caller's plan was refused on the device, nothing to sort */,
               const unsigned lds_cap /* ranges up to this many records (<= kEsLds) are sorted in LDS */) {
    if (skip != nullptr && *skip != 0u) return;
    // (diagnostics, dbg != nullptr: thread 0's time accounts live in LDS -- as registers they cost the whole kernel a dozen VGPRs)
    __shared__ unsigned pf[10];
#define pf_wait pf[0]
#define pf_glob pf[1]
#define pf_lds pf[2]
#define pf_tasks pf[3]
#define pf_t pf[4]
#define pf_lmax pf[5]
#define pf_lmax_m pf[6]
#define pf_gmax pf[7]
#define pf_gmax_m pf[8]
#define pf_m0 pf[9]
#define ES_PF(acc) do { if (dbg && threadIdx.x == 0) { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memrealtime(); acc += now_ - pf_t; pf_t = now_; } } while (0)
    if (dbg && threadIdx.x == 0) { for (int i_ = 0; i_ < 10; ++i_) pf[i_] = 0u; pf_t = (unsigned)__builtin_amdgcn_s_memrealtime(); }
#define ES_MARK(k) do { if (dbg && threadIdx.x == 0 && blockIdx.x == 0 && dbg->mark[k] == 0u) dbg->mark[k] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
    using namespace es_lds;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool take_init = init_n != 0u && blockIdx.x == 0;
    int glvl = 0;  // (diagnostics: partitions out of global memory this workgroup has done)
    for (;;) {
        // ---- pop a task (thread 0), broadcast ----
        if (t == 0 && take_init) {
            int lg = 0;
            while ((2u << lg) <= init_n) ++lg;  // floor(log2 n): introsort's depth limit is twice that
            s_first = 0u; s_last = init_n; s_depth = 2 * lg;
            s_state = 1u;
            s_fail = 0u;
        } else if (t == 0) {
            unsigned state = 0u;  // 1: got a task, 2: all done
            // TICKET queue (round 6): a workgroup draws a slot number ONCE (one atomic add on `head`) and then waits for THAT slot's flag -- a word no other
            // workgroup waits for -- or for the end of all work (`open` == 0: every pushed task has been finished, so slot h will never be filled).
            // Until round 6 every idle workgroup polled {head, tail} and tried a compare-and-swap on `head` whenever a task showed up: with ~200 idle
            // workgroups each push set off ~200 CAS operations on one word, queued in front of the pushing workgroup's own next atomics -- the chain of
            // partitions that feeds the queue ran 3-8x slower than unloaded, and halving the number of workgroups made the kernel 1.5x FASTER
            // (profiles/r06_e_*, r06_f_*).  Slots are handed out in order and filled in order: ticket h gets task h.
            const unsigned h = atomicAdd(&q->head, 1u);
            if (h >= cap) state = 2u;  // (no slot left to wait for: a push beyond the table fails the sort on its own; this workgroup just leaves)
            else {
                const unsigned n_init = q->n_init;
                for (unsigned spin = 0;; ++spin) {
                    if (h < n_init || __hip_atomic_load(&ready[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { state = 1u; break; }
                    if ((spin & 3u) == 3u) {
                        if (es_ld(&q->open) == 0u) {  // nothing unfinished anywhere: slot h stays empty (a task pushed into it would count in `open`)
                            if (__hip_atomic_load(&ready[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { state = 1u; break; }  // (pushed and flagged between the two loads)
                            state = 2u; break;
                        }
                        if ((spin & 63u) == 63u && es_ld(&st->fail)) { state = 2u; break; }
                    }
                    if (spin > 2000000u) { atomicExch(&st->fail, 2u); state = 2u; break; }  // watchdog (seconds): never hang the device
                    if (spin < 8u) __builtin_amdgcn_s_sleep(4); else if (spin < 32u) __builtin_amdgcn_s_sleep(20); else __builtin_amdgcn_s_sleep(60);
                }
                if (state == 1u) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const EsWork wk = tasks[h];
                    s_first = wk.first; s_last = wk.last; s_depth = wk.depth;
                }
            }
            s_state = state;
            s_fail = 0u;
        }
        take_init = false;
        ES_MARK(0);
        __syncthreads();
        ES_MARK(1);
        if (s_state == 2u) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the producer's records, not this CU's cached copies
        ES_MARK(2);
        ES_PF(pf_wait);
        if (dbg && threadIdx.x == 0) { ++pf_tasks; pf_m0 = s_last - s_first; }
        unsigned first = s_first, last = s_last, m = last - first;
        int depth = s_depth;
        bool dead = false;  // the range hit the depth limit (sort failed) -- nothing left to do for this task
        while (m > lds_cap) {
            // ============ a workgroup partitions the range out of global memory, hands one child to the queue and keeps the other:
            // no queue round trip (pop, acquire, ~10 atomics) on the critical path of a lopsided recursion ============
            if (depth == 0) { if (t == 0) atomicExch(&st->fail, 1u); dead = true; break; }  // (introsort switches to heap sort here)
            if (t == 0) atomicAdd(&st->pad[0], 1u);  // diagnostics: partitions done from global memory
#define ES_LVL(j) do { if (dbg && t == 0 && blockIdx.x == 0 && glvl < 32) dbg->lvl[glvl][j] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
            {
                if (dbg && t == 0 && blockIdx.x == 0 && glvl < 32) dbg->lvl[glvl][0] = m;
                ES_LVL(1);
                if (t == 0) s_pivot = es_median_to_first(key, val, first, last);
                __syncthreads();
                ES_LVL(2);
                const unsigned p = s_pivot;
                // wave w owns a contiguous slice of [first + 1, last), a multiple of 64 positions long
                const unsigned span = last - first - 1u, per = ((span + kEsTaskWaves - 1u) / kEsTaskWaves + 63u) & ~63u;
                const unsigned w0 = first + 1u + (unsigned)w * per, w1 = w0 + per < last ? w0 + per : last;
                // (every loop below keeps U independent loads in flight: a global round trip costs ~1 us on a freshly invalidated cache, and a
                // 115 k-record range is 113 rounds of 64 per wave)
                constexpr int U = FLS_ES_UNROLL;  // (8 until the end of round 4: the passes are latency bound, the workgroup is alone on its CU and has registers to spare)
                // (round 6, tried and dropped: the whole slice -- 16 / 32 rounds -- read once and kept in registers from the counts to the stop lists.  16 rounds
                // fit (128 VGPRs, no scratch) and changed nothing (NDT call 0.474-0.479 vs 0.467-0.469 ms), 32 rounds spill: 0.58 ms.  profiles/r06_j_*)
                unsigned cl = 0u, cr = 0u;
                unsigned bl = 0u, br = 0u, nL = 0u, nR = 0u;
                {
                for (unsigned base = w0; base < w1; base += 64u * U) {
                    unsigned kk[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const unsigned i = base + 64u * u + lane; kk[u] = i < w1 ? key[i] : 0u; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned i = base + 64u * u + lane;
                        cl += (unsigned)__popcll(__ballot(i < w1 && kk[u] >= p));
                        cr += (unsigned)__popcll(__ballot(i < w1 && kk[u] <= p));
                    }
                }
                if (lane == 0) { s_wl[w] = cl; s_wr[w] = cr; }
                __syncthreads();
                ES_LVL(3);
                for (int x = 0; x < kEsTaskWaves; ++x) { const unsigned a = s_wl[x], b = s_wr[x]; if (x < w) { bl += a; br += b; } nL += a; nR += b; }
                for (unsigned base = w0; base < w1; base += 64u * U) {
                    unsigned kk[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const unsigned i = base + 64u * u + lane; kk[u] = i < w1 ? key[i] : 0u; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned i = base + 64u * u + lane;
                        const bool fl = i < w1 && kk[u] >= p, fr = i < w1 && kk[u] <= p;
                        const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
                        if (fl) Lp[first + bl + (unsigned)__popcll(ml & lt_mask)] = i;
                        if (fr) Rl[first + br + (unsigned)__popcll(mr & lt_mask)] = i;
                        bl += (unsigned)__popcll(ml); br += (unsigned)__popcll(mr);
                    }
                }
                }
                __syncthreads();
                ES_LVL(4);
                // the K* swaps (cond is monotone in k: K* = the number of k it holds for).  Pair k needs L_k and the k-th R-stop from the
                // right; candidates beyond min(nL, nR + 1) cannot hold.  Conditions first (independent loads), then the records.
                auto Rk = [&](const unsigned k) { return k < nR ? Rl[first + nR - 1u - k] : first; };
                const unsigned kmax = nL < nR + 1u ? nL : nR + 1u;
                unsigned mine = 0u;
                for (unsigned k0 = (unsigned)t; k0 < kmax; k0 += (unsigned)kEsTaskThreads * 4u) {
                    unsigned a[4], b[4];
                    bool c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned k = k0 + (unsigned)u * (unsigned)kEsTaskThreads;
                        a[u] = k < kmax ? Lp[first + k] : 0u;
                        b[u] = k < kmax ? Rk(k) : 0u;
                    }
                    unsigned ka[4], kb[4], va[4], vb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned k = k0 + (unsigned)u * (unsigned)kEsTaskThreads;
                        c[u] = k < kmax && a[u] < b[u];
                        if (c[u]) { ka[u] = key[a[u]]; kb[u] = key[b[u]]; va[u] = val[a[u]]; vb[u] = val[b[u]]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c[u]) { key[a[u]] = kb[u]; key[b[u]] = ka[u]; val[a[u]] = vb[u]; val[b[u]] = va[u]; ++mine; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
                if (lane == 0) s_cnt[w] = mine;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's swaps have left the CU
                __syncthreads();
                ES_LVL(5);
                if (t == 0) {
                    unsigned K = 0u;
                    for (int x = 0; x < kEsTaskWaves; ++x) K += s_cnt[x];
                    const unsigned INF = 0xFFFFFFFFu;
                    const unsigned a = K < nL ? Lp[first + K] : INF;
                    const unsigned b = K >= 1u ? Rk(K - 1u) : INF;
                    const unsigned cut = a < b ? a : b;
                    // keep the larger child, publish the other (its records: every wave's swaps reached the L2 before the barrier above; the
                    // release of es_push writes them back for a consumer on another XCD)
                    const bool keep_right = last - cut >= cut - first;
                    if (keep_right) { es_push(q, tasks, ready, cap, st, first, cut, depth - 1); s_first = cut; s_last = last; }
                    else { es_push(q, tasks, ready, cap, st, cut, last, depth - 1); s_first = first; s_last = cut; }
                }
                __syncthreads();
                ES_LVL(6);
                if (blockIdx.x == 0) ++glvl;
                first = s_first; last = s_last; m = last - first; --depth;
            }
        }
        ES_MARK(9);
        if (dbg && threadIdx.x == 0) { const unsigned before_ = pf_glob; ES_PF(pf_glob); if (pf_glob - before_ > pf_gmax) { pf_gmax = pf_glob - before_; pf_gmax_m = pf_m0; } }
        if (!dead && m >= 2u) {
            if (t == 0) { atomicAdd(&st->pad[1], 1u); atomicAdd(&st->pad[2], m); }  // diagnostics: ranges sorted in LDS, their records
            // ============ the range lives in LDS; its waves partition sub-ranges from a local queue ============
            for (unsigned i = t; i < m; i += kEsTaskThreads) { sk[i] = key[first + i]; sv[i] = val[first + i]; }
            for (unsigned i = t; i < (m >> 5) + 2u; i += kEsTaskThreads) bmask[i] = 0u;  // (the words records 0 .. m can touch)
            __syncthreads();
            for (unsigned i = t; i < (unsigned)kEsLocalQ; i += kEsTaskThreads) qt[i] = 0u;
            if (t == 0) {
                atomicOr(&bmask[0], 1u);
                atomicOr(&bmask[m >> 5], 1u << (m & 31u));
                lq_head = 0u; lq_tail = 0u; lq_open = 0u;
                s_sp = 0u;
                if (m > (unsigned)kEsCoop) { sa_f[0] = 0; sa_l[0] = (unsigned short)m; sa_d[0] = (signed char)(depth > 127 ? 127 : depth); s_sp = 1u; }
                else if (m > (unsigned)kEsThreshold) { qd[0] = (signed char)(depth > 127 ? 127 : depth); lq_tail = 1u; lq_open = 1u; }
            }
            __syncthreads();
            if (t == 0 && m > (unsigned)kEsThreshold && m <= (unsigned)kEsCoop) qt[0] = m << 16;  // (first = 0)
            __syncthreads();
            ES_MARK(10);
            // ---- phase A: sub-ranges longer than kEsCoop are partitioned by the WHOLE workgroup, one after the other (a stack): a
            // lopsided recursion keeps one long sub-range alive for many levels, and a single wave needs m / 64 dependent rounds for it
            while (s_sp != 0u && !s_fail) {
                const unsigned sp = s_sp - 1u;
                const unsigned f = sa_f[sp], l = sa_l[sp];
                const int d = sa_d[sp];
                __syncthreads();  // (everyone has read the top of the stack)
                if (d == 0) {  // depth limit: heap sort (one lane), every record of the range becomes a final block of its own, next entry
                    if (t == 0) es_heap_sort(sk + f, sv + f, (int)(l - f));
                    __syncthreads();
                    for (unsigned i = f + 1u + (unsigned)t; i < l; i += (unsigned)kEsTaskThreads) atomicOr(&bmask[i >> 5], 1u << (i & 31u));
                    if (t == 0) s_sp = sp;
                    __syncthreads();
                    continue;
                }
                if (t == 0) {
                    const unsigned a = f + 1u, b = f + (l - f) / 2u, c = l - 1u;
                    const unsigned ka = sk[a], kb = sk[b], kc = sk[c];
                    unsigned med;
                    if (ka < kb) { if (kb < kc) med = b; else if (ka < kc) med = c; else med = a; }
                    else if (ka < kc) med = a;
                    else if (kb < kc) med = c;
                    else med = b;
                    const unsigned k0 = sk[f], v0 = sv[f];
                    sk[f] = sk[med]; sv[f] = sv[med]; sk[med] = k0; sv[med] = v0;
                    s_pivot = sk[f];
                }
                __syncthreads();
                const unsigned p = s_pivot;
                const unsigned span = l - f - 1u, per = ((span + kEsTaskWaves - 1u) / kEsTaskWaves + 63u) & ~63u;
                const unsigned w0 = f + 1u + (unsigned)w * per, w1 = w0 + per < l ? w0 + per : l;
                unsigned cl = 0u, cr = 0u;
                for (unsigned base = w0; base < w1; base += 64u) {
                    const unsigned i = base + lane;
                    const unsigned k = i < w1 ? sk[i] : 0u;
                    cl += (unsigned)__popcll(__ballot(i < w1 && k >= p));
                    cr += (unsigned)__popcll(__ballot(i < w1 && k <= p));
                }
                if (lane == 0) { s_wl[w] = cl; s_wr[w] = cr; }
                __syncthreads();
                unsigned bl = 0u, br = 0u, nL = 0u, nR = 0u;
                for (int x = 0; x < kEsTaskWaves; ++x) { const unsigned a = s_wl[x], b = s_wr[x]; if (x < w) { bl += a; br += b; } nL += a; nR += b; }
                for (unsigned base = w0; base < w1; base += 64u) {
                    const unsigned i = base + lane;
                    const unsigned k = i < w1 ? sk[i] : 0u;
                    const bool fl = i < w1 && k >= p, fr = i < w1 && k <= p;
                    const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
                    if (fl) lp[f + bl + (unsigned)__popcll(ml & lt_mask)] = (unsigned short)i;
                    if (fr) rl[f + br + (unsigned)__popcll(mr & lt_mask)] = (unsigned short)i;
                    bl += (unsigned)__popcll(ml); br += (unsigned)__popcll(mr);
                }
                __syncthreads();
                const unsigned kmax = nL < nR + 1u ? nL : nR + 1u;
                unsigned mine = 0u;
                for (unsigned k = (unsigned)t; k < kmax; k += (unsigned)kEsTaskThreads) {
                    const unsigned a = lp[f + k], b = k < nR ? (unsigned)rl[f + nR - 1u - k] : f;
                    if (a < b) { const unsigned ka = sk[a], va = sv[a]; sk[a] = sk[b]; sv[a] = sv[b]; sk[b] = ka; sv[b] = va; ++mine; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
                if (lane == 0) s_cnt[w] = mine;
                __syncthreads();
                if (t == 0) {
                    unsigned K = 0u;
                    for (int x = 0; x < kEsTaskWaves; ++x) K += s_cnt[x];
                    const unsigned INF = 0xFFFFFFFFu;
                    const unsigned a = K < nL ? (unsigned)lp[f + K] : INF;
                    const unsigned b = K >= 1u ? (K - 1u < nR ? (unsigned)rl[f + nR - K] : f) : INF;
                    const unsigned cut = a < b ? a : b;
                    atomicOr(&bmask[cut >> 5], 1u << (cut & 31u));
                    unsigned top = sp;  // (the processed entry is replaced)
                    const unsigned cf[2] = {f, cut}, cl2[2] = {cut, l};
                    for (int x = 0; x < 2; ++x) {
                        const unsigned cm = cl2[x] - cf[x];
                        if (cm > (unsigned)kEsCoop) {
                            if (top < (unsigned)kEsStack) { sa_f[top] = (unsigned short)cf[x]; sa_l[top] = (unsigned short)cl2[x]; sa_d[top] = (signed char)(d - 1); ++top; }
                            else s_fail = 1u;
                        } else if (cm > (unsigned)kEsThreshold) {
                            const unsigned s2 = lq_tail;
                            if (s2 < (unsigned)kEsLocalQ) { qd[s2] = (signed char)(d - 1); qt[s2] = cf[x] | (cl2[x] << 16); lq_tail = s2 + 1u; lq_open = lq_open + 1u; }
                            else s_fail = 1u;
                        }
                    }
                    s_sp = top;
                }
                __syncthreads();
            }
            ES_MARK(3);
            // ---- phase B (round 5): the remaining sub-ranges (<= kEsCoop records) are tasks of a TICKET QUEUE in LDS.  A wave takes a ticket,
            // waits until that slot is published (or until no task is open any more: nothing will ever be published), partitions the range
            // and keeps the SMALLER child on its private stack; a larger child longer than kEsShare records goes back to the queue, where an
            // idle wave picks it up.  (Round 4 gave every queued range with its whole subtree to one wave: <= 8 tasks for 16 waves, 40-80 us
            // on the slowest wave -- profiles/r05_a_vg_*_before_exact_sort_stamps.log.)  open = queue tasks not yet finished, subtrees included.
            es_phase_b(m);
            ES_MARK(4);
            __syncthreads();
            ES_MARK(5);
            if (s_fail) { if (t == 0) atomicExch(&st->fail, 1u); }
            else {
                // the insertion sort: every record moves to its stable rank inside its final block (blocks of <= 16 between cuts)
                for (unsigned i = t; i < m; i += kEsTaskThreads) {
                    // the record's final block [b0, b1): the nearest cut at or below i, the nearest above (blocks hold <= 16 records, so
                    // both lie within two mask words)
                    unsigned b0, b1;
                    {
                        const unsigned wi = i >> 5, bi = i & 31u;
                        const unsigned lo = bmask[wi] & (0xFFFFFFFFu >> (31u - bi));  // bits 0 .. bi
                        if (lo) b0 = (wi << 5) + (31u - (unsigned)__clz((int)lo));
                        else { const unsigned pw = bmask[wi - 1u]; b0 = ((wi - 1u) << 5) + (31u - (unsigned)__clz((int)pw)); }
                        const unsigned hi = bi == 31u ? 0u : (bmask[wi] & (0xFFFFFFFFu << (bi + 1u)));  // bits bi + 1 .. 31
                        if (hi) b1 = (wi << 5) + (unsigned)__ffs((int)hi) - 1u;
                        else { const unsigned nw = bmask[wi + 1u]; b1 = ((wi + 1u) << 5) + (unsigned)__ffs((int)nw) - 1u; }
                    }
                    const unsigned k = sk[i];
                    unsigned r = 0u;
                    for (unsigned j = b0; j < b1; ++j) { const unsigned kj = sk[j]; r += (kj < k || (kj == k && j < i)) ? 1u : 0u; }
                    key[first + b0 + r] = k;
                    val[first + b0 + r] = sv[i];
                }
            }
        }
        ES_MARK(6);
        __syncthreads();
        if (t == 0) atomicSub(&q->open, 1u);  // (after the children were pushed)
        ES_MARK(7);
        if (dbg && threadIdx.x == 0) { const unsigned before_ = pf_lds; ES_PF(pf_lds); if (pf_lds - before_ > pf_lmax) { pf_lmax = pf_lds - before_; pf_lmax_m = m; } }
    }
    ES_MARK(8);
    if (dbg && threadIdx.x == 0 && blockIdx.x < 256u) {
        ES_PF(pf_wait);
        dbg->wg[blockIdx.x][0] = pf_wait; dbg->wg[blockIdx.x][1] = pf_glob; dbg->wg[blockIdx.x][2] = pf_lds; dbg->wg[blockIdx.x][3] = pf_tasks;
        dbg->wg[blockIdx.x][4] = pf_lmax; dbg->wg[blockIdx.x][5] = pf_lmax_m; dbg->wg[blockIdx.x][6] = pf_gmax; dbg->wg[blockIdx.x][7] = pf_gmax_m;
    }
#undef ES_PF
#undef pf_wait
#undef pf_glob
#undef pf_lds
#undef pf_tasks
#undef pf_t
#undef pf_lmax
#undef pf_lmax_m
#undef pf_gmax
#undef pf_gmax_m
#undef pf_m0
#undef ES_MARK
}

}  // namespace fls
