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
//   regime 1  ranges longer than kEsLds records: four launches per level over all such ranges (es_level_begin: cuts + children of the
//             previous level, medians of this one; es_count / es_scatter: stop lists through per-tile counts; es_swap)
//   regime 2  every range of <= kEsLds records: one workgroup takes it into LDS and runs ALL its remaining levels there
//             (es_lds_kernel), then ranks the records of every final <= 16 block (= the insertion sort) and writes them back.
// The heap-sort fallback of introsort (recursion deeper than 2 log2 n: adversarial inputs) is not reproduced: the sort reports
// failure and the caller takes the host path.
#pragma once
#include "device_common.hpp"

namespace fls {

constexpr int kEsLds = 4096;        // records a workgroup sorts in LDS
constexpr int kEsLdsThreads = 512;
constexpr int kEsLdsItems = kEsLds / kEsLdsThreads;
constexpr int kEsMaxSub = 256;      // active sub-ranges of one level inside an LDS range (<= kEsLds / 17)
constexpr int kEsThreshold = 16;    // _S_threshold
constexpr int kEsTile = 2048, kEsBlock = 256, kEsItems = kEsTile / kEsBlock;
constexpr int kEsMaxSeg = 2048;     // regime-1 ranges of one level (n / kEsLds for n <= 4 Mi, with room)

struct EsSeg { unsigned first, last; int depth; unsigned pivot, nL, nR, K, tile0; };
struct EsWork { unsigned first, last; int depth, pad; };
struct EsState {
    unsigned n_cur;     // regime-1 ranges of the level in flight
    unsigned n_tiles;   // their tiles
    unsigned n_work;    // ranges handed to regime 2 so far
    unsigned fail;      // 1: introsort would heap-sort / a table overflowed -> the caller takes the host path
    unsigned level;
    unsigned pad[3];
};
// what the host polls (host-mapped): written by every es_level_begin
struct EsMailbox { unsigned seq, n_cur, n_work, fail; };

__device__ __forceinline__ void es_swap_rec(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned a, const unsigned b) {
    const unsigned ka = key[a], kb = key[b], va = val[a], vb = val[b];
    key[a] = kb; key[b] = ka; val[a] = vb; val[b] = va;
}
// std::__move_median_to_first(result = first, a = first + 1, b = mid, c = last - 1) on the keys; returns the pivot key
__device__ __forceinline__ unsigned es_median_to_first(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned first, const unsigned last) {
    const unsigned a = first + 1u, b = first + (last - first) / 2u, c = last - 1u;
    const unsigned ka = key[a], kb = key[b], kc = key[c];
    unsigned med;
    if (ka < kb) { if (kb < kc) med = b; else if (ka < kc) med = c; else med = a; }
    else if (ka < kc) med = a;
    else if (kb < kc) med = c;
    else med = b;
    es_swap_rec(key, val, first, med);
    return key[first];
}

// ---- regime 1 ----------------------------------------------------------------------------------------------------------------------
// One workgroup: finish the previous level (cut of every range from its K*, children -> next level or the LDS work list), then open
// the new level (median of three, pivot, tile map).  `prev` / `cur` are the two range tables, swapped by the host every level.
__global__ void __launch_bounds__(256)
es_level_begin(unsigned* __restrict__ key, unsigned* __restrict__ val, const unsigned n, const EsSeg* __restrict__ prev, EsSeg* __restrict__ cur,
               EsWork* __restrict__ work, const unsigned work_cap, const unsigned* __restrict__ Lp, const unsigned* __restrict__ Rl,
               EsState* __restrict__ st, EsMailbox* __restrict__ mb, const unsigned seq, const int first_level, unsigned* __restrict__ tile_seg,
               const unsigned tile_cap) {
    __shared__ unsigned s_ncur, s_nwork, s_fail, s_tiles[kEsMaxSeg], s_total;
    if (threadIdx.x == 0) { s_ncur = 0u; s_nwork = first_level ? 0u : st->n_work; s_fail = first_level ? 0u : st->fail; }
    __syncthreads();
    auto child = [&](const unsigned f, const unsigned l, const int depth) {
        const unsigned m = l - f;
        if (m < 2u) return;
        if (m > (unsigned)kEsLds) {
            const unsigned slot = atomicAdd(&s_ncur, 1u);
            if (slot < (unsigned)kEsMaxSeg) cur[slot] = EsSeg{f, l, depth, 0u, 0u, 0u, 0u, 0u};
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
        EsSeg g = cur[s];
        if (g.depth == 0) { s_fail = 1u; g.pivot = key[g.first]; }  // (introsort switches to heap sort here)
        else g.pivot = es_median_to_first(key, val, g.first, g.last);
        cur[s] = g;
    }
    __syncthreads();
    for (unsigned s = threadIdx.x; s < nc; s += 256u) s_tiles[s] = (cur[s].last - cur[s].first - 1u + (unsigned)kEsTile - 1u) / (unsigned)kEsTile;
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix of the tile counts (LDS, <= kEsMaxSeg entries)
        unsigned t = 0u;
        for (unsigned s = 0; s < nc; ++s) { const unsigned c = s_tiles[s]; s_tiles[s] = t; t += c; }
        s_total = t;
        if (t > tile_cap) s_fail = 1u;
    }
    __syncthreads();
    if (!s_fail)
        for (unsigned s = threadIdx.x; s < nc; s += 256u) {  // tile -> range map (one load per workgroup in the three launches that follow)
            const unsigned t0 = s_tiles[s], t1 = s + 1u < nc ? s_tiles[s + 1u] : s_total;
            cur[s].tile0 = t0;
            for (unsigned q = t0; q < t1; ++q) tile_seg[q] = s;
        }
    if (threadIdx.x == 0) {
        const unsigned t = s_total;
        st->n_cur = s_fail ? 0u : nc;
        st->n_tiles = s_fail ? 0u : t;
        st->n_work = s_nwork < work_cap ? s_nwork : work_cap;
        st->fail = s_fail;
        st->level = first_level ? 0u : st->level + 1u;
        __hip_atomic_store(&mb->n_cur, st->n_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->n_work, st->n_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->fail, s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
// the K* swaps of every range (pair k = L_k <-> R_k, disjoint), and K* itself (the one k with cond(k) && !cond(k + 1))
__global__ void __launch_bounds__(kEsBlock)
es_swap_kernel(unsigned* __restrict__ key, unsigned* __restrict__ val, EsSeg* __restrict__ cur, const EsState* __restrict__ st,
               const unsigned* __restrict__ tile_seg, const unsigned* __restrict__ Lp, const unsigned* __restrict__ Rl) {
    unsigned seg, tile;
    if (!es_locate(cur, st, tile_seg, seg, tile)) return;
    const EsSeg g = cur[seg];
    auto Lk = [&](const unsigned k) { return Lp[g.first + k]; };
    auto Rk = [&](const unsigned k) { return k < g.nR ? Rl[g.first + g.nR - 1u - k] : g.first; };  // (the pivot itself stops the downward scan last)
    auto cond = [&](const unsigned k) { return k < g.nL && Lk(k) < Rk(k); };
#pragma unroll
    for (int q = 0; q < kEsItems; ++q) {
        const unsigned k = tile * (unsigned)kEsTile + q * (unsigned)kEsBlock + threadIdx.x;
        if (k >= g.last - g.first - 1u) continue;
        if (!cond(k)) continue;
        es_swap_rec(key, val, Lk(k), Rk(k));
        if (!cond(k + 1u)) cur[seg].K = k + 1u;
    }
}

// ---- regime 2: one workgroup per range of <= kEsLds records, every remaining level in LDS -----------------------------------------
__global__ void __launch_bounds__(kEsLdsThreads)
es_lds_kernel(unsigned* __restrict__ key, unsigned* __restrict__ val, const EsWork* __restrict__ work, EsState* __restrict__ st) {
    __shared__ unsigned sk[kEsLds], sv[kEsLds];
    __shared__ unsigned short lp[kEsLds], rl[kEsLds], pl[kEsLds + 1], pr[kEsLds + 1], sid[kEsLds];
    __shared__ unsigned char bnd[kEsLds + 1];
    __shared__ unsigned short sf[2][kEsMaxSub], sl[2][kEsMaxSub], sK[kEsMaxSub], ida[kEsMaxSub], idb[kEsMaxSub], scut[kEsMaxSub];
    __shared__ short sd[2][kEsMaxSub];
    __shared__ unsigned spiv[kEsMaxSub];
    __shared__ unsigned wsum[kEsLdsThreads / 64][2];
    __shared__ unsigned s_nact, s_nnext, s_fail;
    const int t = threadIdx.x;
    const unsigned n_work = st->n_work;
    for (unsigned w = blockIdx.x; w < n_work; w += gridDim.x) {
        const EsWork wk = work[w];
        const unsigned m = wk.last - wk.first;
        __syncthreads();  // (LDS of the previous range is free)
        for (unsigned i = t; i < m; i += kEsLdsThreads) { sk[i] = key[wk.first + i]; sv[i] = val[wk.first + i]; sid[i] = m > (unsigned)kEsThreshold ? 0 : 0xffff; bnd[i] = 0; }
        if (t == 0) {
            bnd[0] = 1; bnd[m] = 1;
            s_nact = m > (unsigned)kEsThreshold ? 1u : 0u;
            s_fail = 0u;
            sf[0][0] = 0; sl[0][0] = (unsigned short)m; sd[0][0] = (short)wk.depth;
        }
        __syncthreads();
        int cur = 0;
        while (s_nact != 0u && !s_fail) {
            const unsigned nact = s_nact;
            // A: median of three per active sub-range
            for (unsigned s = t; s < nact; s += kEsLdsThreads) {
                const unsigned f = sf[cur][s], l = sl[cur][s];
                if (sd[cur][s] == 0) { s_fail = 1u; continue; }
                const unsigned a = f + 1u, b = f + (l - f) / 2u, c = l - 1u;
                const unsigned ka = sk[a], kb = sk[b], kc = sk[c];
                unsigned med;
                if (ka < kb) { if (kb < kc) med = b; else if (ka < kc) med = c; else med = a; }
                else if (ka < kc) med = a;
                else if (kb < kc) med = c;
                else med = b;
                const unsigned k0 = sk[f], v0 = sv[f];
                sk[f] = sk[med]; sv[f] = sv[med]; sk[med] = k0; sv[med] = v0;
                spiv[s] = sk[f];
                sK[s] = 0;
            }
            if (t == 0) s_nnext = 0u;
            __syncthreads();
            if (s_fail) break;
            // B: both predicates of every record of an active sub-range (not its pivot slot), ranked by one block-wide exclusive scan
            unsigned fl[kEsLdsItems], fr[kEsLdsItems];
            unsigned v[2] = {0u, 0u}, tot[2];
#pragma unroll
            for (int q = 0; q < kEsLdsItems; ++q) {
                const unsigned i = (unsigned)t * kEsLdsItems + q;
                fl[q] = fr[q] = 0u;
                if (i < m) {
                    const unsigned s = sid[i];
                    if (s != 0xffffu && i != sf[cur][s]) { const unsigned k = sk[i], p = spiv[s]; fl[q] = k >= p ? 1u : 0u; fr[q] = k <= p ? 1u : 0u; }
                }
                v[0] += fl[q]; v[1] += fr[q];
            }
            block_excl_scan<2>(v, tot, wsum);
            {
                unsigned a = v[0], b = v[1];
#pragma unroll
                for (int q = 0; q < kEsLdsItems; ++q) {
                    const unsigned i = (unsigned)t * kEsLdsItems + q;
                    if (i < m) { pl[i] = (unsigned short)a; pr[i] = (unsigned short)b; }
                    a += fl[q]; b += fr[q];
                }
                if (t == 0) { pl[m] = (unsigned short)tot[0]; pr[m] = (unsigned short)tot[1]; }
            }
            __syncthreads();
            // C: stop lists of every sub-range at [first + rank]
#pragma unroll
            for (int q = 0; q < kEsLdsItems; ++q) {
                const unsigned i = (unsigned)t * kEsLdsItems + q;
                if (i < m && (fl[q] | fr[q])) {
                    const unsigned f = sf[cur][sid[i]];
                    if (fl[q]) lp[f + pl[i] - pl[f + 1u]] = (unsigned short)i;
                    if (fr[q]) rl[f + pr[i] - pr[f + 1u]] = (unsigned short)i;
                }
            }
            __syncthreads();
            // D: swaps (position i of a sub-range plays k = i - first) and K*
            for (unsigned i = t; i < m; i += kEsLdsThreads) {
                const unsigned s = sid[i];
                if (s == 0xffffu) continue;
                const unsigned f = sf[cur][s], l = sl[cur][s], k = i - f;
                if (k >= l - f - 1u) continue;
                const unsigned nL = (unsigned)pl[l] - pl[f + 1u], nR = (unsigned)pr[l] - pr[f + 1u];
                auto Rk = [&](const unsigned kq) { return kq < nR ? (unsigned)rl[f + nR - 1u - kq] : f; };
                auto cond = [&](const unsigned kq) { return kq < nL && (unsigned)lp[f + kq] < Rk(kq); };
                if (!cond(k)) continue;
                const unsigned a = lp[f + k], b = Rk(k);
                const unsigned ka = sk[a], va = sv[a];
                sk[a] = sk[b]; sv[a] = sv[b]; sk[b] = ka; sv[b] = va;
                if (!cond(k + 1u)) sK[s] = (unsigned short)(k + 1u);
            }
            __syncthreads();
            // E: cuts, children, the next level's table
            for (unsigned s = t; s < nact; s += kEsLdsThreads) {
                const unsigned f = sf[cur][s], l = sl[cur][s], K = sK[s];
                const unsigned nL = (unsigned)pl[l] - pl[f + 1u], nR = (unsigned)pr[l] - pr[f + 1u];
                const unsigned INF = 0xFFFFFFFFu;
                const unsigned a = K < nL ? (unsigned)lp[f + K] : INF;
                const unsigned b = K >= 1u ? (K - 1u < nR ? (unsigned)rl[f + nR - K] : f) : INF;
                const unsigned cut = a < b ? a : b;
                scut[s] = (unsigned short)cut;
                bnd[cut] = 1;
                const short d = (short)(sd[cur][s] - 1);
                unsigned short ia = 0xffff, ib = 0xffff;
                if (cut - f > (unsigned)kEsThreshold) {
                    const unsigned slot = atomicAdd(&s_nnext, 1u);
                    if (slot < (unsigned)kEsMaxSub) { sf[cur ^ 1][slot] = (unsigned short)f; sl[cur ^ 1][slot] = (unsigned short)cut; sd[cur ^ 1][slot] = d; ia = (unsigned short)slot; }
                    else s_fail = 1u;
                }
                if (l - cut > (unsigned)kEsThreshold) {
                    const unsigned slot = atomicAdd(&s_nnext, 1u);
                    if (slot < (unsigned)kEsMaxSub) { sf[cur ^ 1][slot] = (unsigned short)cut; sl[cur ^ 1][slot] = (unsigned short)l; sd[cur ^ 1][slot] = d; ib = (unsigned short)slot; }
                    else s_fail = 1u;
                }
                ida[s] = ia; idb[s] = ib;
            }
            __syncthreads();
            for (unsigned i = t; i < m; i += kEsLdsThreads) {
                const unsigned s = sid[i];
                if (s != 0xffffu) sid[i] = i < scut[s] ? ida[s] : idb[s];
            }
            if (t == 0) s_nact = s_nnext;
            cur ^= 1;
            __syncthreads();
        }
        if (s_fail) { if (t == 0) st->fail = 1u; continue; }
        // the insertion sort: every record moves to its stable rank inside its final block (blocks of <= 16 between cuts)
        for (unsigned i = t; i < m; i += kEsLdsThreads) {
            unsigned b0 = i, b1 = i + 1u;
            while (!bnd[b0]) --b0;
            while (!bnd[b1]) ++b1;
            const unsigned k = sk[i];
            unsigned r = 0u;
            for (unsigned j = b0; j < b1; ++j) { const unsigned kj = sk[j]; r += (kj < k || (kj == k && j < i)) ? 1u : 0u; }
            key[wk.first + b0 + r] = k;
            val[wk.first + b0 + r] = sv[i];
        }
    }
}

}  // namespace fls
