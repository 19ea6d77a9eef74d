// kernels_ivox_update.hpp -- IVoxMap::AddPoints (src/ivox_map/ivox_map.cpp:122-143) on the device, for the map update that
// LoamPointToPlaneIVOX::Match performs in mapping mode (loam_point_to_plane_ivox.h:60-139, 205-206).  SURVEY.md 8f rank 1.
//
// The reference inserts sequentially: first every "points_to_add" point in source order, then every
// "point_no_need_downsample" point in source order; a point either creates its voxel at the LRU front or is appended to it
// (and moves it to the front); a creation that brings the voxel count to the capacity evicts the LRU tail.
// As long as no eviction happens inside a batch, the final state is a function of the per-point SEQUENCE RANK only:
//   * a point's insertion id            = next_id + rank
//   * the order of points in a voxel    = rank order
//   * the LRU order of the voxels       = order of the rank of their LAST inserted point (older voxels keep theirs): a stamp
//   * where a grown / new voxel's slot region goes = order of the rank of its FIRST inserted point (the order in which the
//     round-1 host mirror relocated regions, kept so that the image -- and every exact-tie decision of the kNN kernel
//     -- is the one the host path builds)
// so the batch is applied with prefix sums over the sequence and per-voxel atomics whose arbitrary arrival order is erased again
// (min / max / count are order free; the slots a voxel's new points land in are sorted by id afterwards).  Deterministic.
// LRU evictions inside the batch (voxel count reaching the capacity) are selected by a walk over the alive cells in stamp order
// (ivox_evict_select below), including voxels that are evicted and re-created by a later point of the same batch.
// Anything this path cannot do exactly -- evictions that would reach voxels the batch itself created or touched, a key beyond
// +-2^20, the point array or the brick pool running out of room -- is detected BEFORE any map state is touched; the batch is
// then not applied, the status word says so and the host takes the exact sequential path (matcher_p2plane_ivox.hpp).
//
// Launch sequence (one stream, no host round trip; n = source points, A = points to insert):
//   ivox_upd_count      <n>   per-block counts of the two decision codes
//   ivox_upd_scan1      <1>   block offsets, totals
//   ivox_upd_seq        <n>   rank, voxel cell, sequence arrays; pending count / first rank per cell (atomics)
//   ivox_upd_plan       <A>   first-toucher flags; per touched voxel: new total, region growth, creation -> block scans
//   ivox_upd_scan2      <1>   block offsets, totals, the all-or-nothing checks
//   ivox_upd_last       <A>   first rank -> last rank per cell (atomicMax on the same word)
//   ivox_upd_regions    <A>   per touched voxel: relocate if grown, new {begin, count}, capacity, touched list
//   ivox_upd_points     <A>   write the new points
//   ivox_upd_finish     <T waves>  per touched voxel: order its new points by id (rank count in LDS), stamp, reset the temporaries
//   ivox_upd_commit     <1>   counters, status to the host-mapped word
#pragma once
#include "device_common.hpp"
#include "kernels_p2plane.hpp"  // fanin_last_arriver

namespace fls {

constexpr unsigned kUpdInvalidCell = 0xFFFFFFFFu;
constexpr unsigned kUpdNoRank = 0xFFFFFFFFu;
constexpr int kUpdBlock = 256;
constexpr int kUpdMaxBlocks = 1024;  // one-workgroup scan of the block totals: n <= 262,144 source points per batch

enum : unsigned { kUpdOk = 0u, kUpdNeedHost = 1u, kUpdEvictConflict = 2u, kUpdArrayFull = 4u, kUpdOutside = 8u,  // any bit set: the batch is refused
                  kUpdSkipped = 16u };  // a speculative chain found that the Match does not update the map here: nothing ran (ivox_add_decide_kernel)

// persistent device-side bookkeeping of the map image + the per-batch scratch words
struct IvoxUpdState {
    unsigned long long n_points, used, garbage, stamp_base, pts_capacity;
    unsigned n_alive, lru_capacity;
    int next_id;
    unsigned status;          // of the batch in flight: kUpdOk / kUpdNeedHost (sticky inside one batch)
    unsigned n1, n2;          // batch: points of code 1 / code 2
    unsigned alloc, creations, touched, relocated_garbage;  // batch totals (ivox_upd_scan2)
    unsigned apply;           // 1: the batch passes every check and is applied
    unsigned evict;           // voxels this batch evicts (LRU capacity reached inside the batch)
    unsigned evict_ready, n_list;  // the host queued the eviction selection; entries of the stamp-sorted list of alive cells
    unsigned long long evicted_points, evicted_slots;  // totals of the evicted voxels (ivox_evict_apply)
    unsigned n_bricks;        // bricks in the directory (device-side creation: ivox_upd_seq); may overshoot the pool inside a refused batch, clamped at commit
    unsigned recreated;       // voxels this batch evicts AND re-creates (touched after their turn in the eviction order: ivox_evict_select)
};
// what the host reads back (host-mapped pinned memory, written by ivox_upd_commit)
struct IvoxUpdMailbox {
    unsigned long long n_points, used, garbage;
    unsigned n_alive, status, added, touched;
    int next_id;
    unsigned seq;
    unsigned evicted, n_bricks;
    unsigned recreated, pad_;
};

struct IvoxUpdArrays {
    uint2* cells;            // {begin, count} per window cell (the image the kNN kernel reads)
    float4* pts;
    unsigned char* cap_log2;  // per cell: log2 of its slot region's capacity (0: no region)
    unsigned long long* stamp;  // per cell: LRU stamp of the last insertion (larger = more recent); 64-bit, never wraps
    unsigned* pend;          // per cell scratch: points of this batch (0 between batches)
    unsigned* rank_mm;       // per cell scratch: first, later last, rank of this batch (kUpdNoRank between batches)
    HashEntry* dir;          // brick directory {packed brick key, brick index} (device_common.hpp BrickDir), open addressing
    unsigned dir_mask;
    unsigned n_bricks_cap;   // brick pool size: slabs [0, n_bricks_cap) x kBrickStride cells exist (zero beyond the bricks in use)
    unsigned long long* brick_key;  // [n_bricks_cap] packed key of brick i
    unsigned* nbr;           // [n_bricks_cap][32] device-private cache: index of the neighbouring brick with code (dx+1) + 3 (dy+1) + 9 (dz+1), kBrickPending = not looked up yet
                             // (write-once values: a stale "not yet" only sends the reader through the directory again)
    float inv_res;
};
__device__ __forceinline__ int brick_nbr_code(const int dx, const int dy, const int dz) { return (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1); }

// ---- brick directory on the device ----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned brick_find(const IvoxUpdArrays& a, const int bx, const int by, const int bz) {
    const unsigned long long key = pack_key(bx, by, bz);
    for (unsigned h = brick_hash(bx, by, bz) & a.dir_mask;; h = (h + 1u) & a.dir_mask) {
        const unsigned long long k = a.dir[h].key;
        if (k == key) return a.dir[h].begin;
        if (k == kEmptyKey) return kBrickInvalid;
    }
}
// Find the brick or create it (ivox_upd_seq only: the one kernel that may add bricks).  A creation claims an empty directory entry with
// a 64-bit CAS on its key, draws the next slab of the pre-zeroed pool and publishes the index; a thread that meets a claimed entry whose
// index is still pending retries -- the winner publishes inside the SAME loop iteration it won in, so lanes of one wave cannot wait on
// each other.  A new brick is empty (all cells zero) whatever happens to the batch: creating it early is harmless if the batch is refused.
// Pool or directory exhausted: the entry gets kBrickInvalid, the batch is refused (kUpdArrayFull) and the host rebuilds with more room.
__device__ __forceinline__ unsigned brick_find_or_create(const IvoxUpdArrays& a, IvoxUpdState* __restrict__ st, const int bx, const int by, const int bz) {
    const unsigned long long key = pack_key(bx, by, bz);
    unsigned h = brick_hash(bx, by, bz) & a.dir_mask;
    {   // fast path: the entry is there and published (one plain 16-byte load; entries are write-once, so a stale line can only look
        // emptier than the truth and sends us to the careful loop below)
        const HashEntry e = a.dir[h];
        if (e.key == key && e.begin < a.n_bricks_cap) return e.begin;  // (an invalid index falls through to the careful loop, which raises kUpdArrayFull)
    }
    for (unsigned tries = 0; tries < 65536u; ++tries) {
        const unsigned long long k = __hip_atomic_load(&a.dir[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == key) {
            const unsigned idx = __hip_atomic_load(&a.dir[h].begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx != kBrickPending) {
                // an index beyond the pool (kBrickInvalid, published by the claimant of a full pool) must never coexist with an OK verdict,
                // whoever reads it and in whichever batch (today the host rebuilds the directory before the next batch anyway)
                if (idx >= a.n_bricks_cap) atomicOr(&st->status, kUpdArrayFull);
                return idx;
            }
            continue;  // claimed by another thread, index not yet published
        }
        if (k == kEmptyKey) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&a.dir[h].key, kEmptyKey, key);
            if (prev == kEmptyKey) {
                unsigned idx = atomicAdd(&st->n_bricks, 1u);
                if (idx >= a.n_bricks_cap) { idx = kBrickInvalid; atomicOr(&st->status, kUpdArrayFull); }
                else a.brick_key[idx] = key;
                __hip_atomic_store(&a.dir[h].begin, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return idx;
            }
            if (prev == key) continue;  // lost the race for the same brick: read its index
        }
        h = (h + 1u) & a.dir_mask;
    }
    atomicOr(&st->status, kUpdArrayFull);
    return kBrickInvalid;
}
// the halo copies of primary cell `cell` (a boundary voxel of its brick) in the neighbouring bricks' slabs; the neighbours' indices come
// from the brick's cache row (ivox_upd_seq filled it when the voxel's bricks were ensured), the directory is only the fallback
__device__ __forceinline__ void brick_write_mirrors(const IvoxUpdArrays& a, const unsigned cell, const uint2 val) {
    int sx, sy, sz;
    if (!brick_slab_interior(cell & (kBrickStride - 1u), sx, sy, sz)) return;
    const unsigned bi = cell / kBrickStride;
    const unsigned* const row = a.nbr + (size_t)bi * 32u;
    brick_for_each_mirror(sx - 1, sy - 1, sz - 1, [&](const int dx, const int dy, const int dz) {
        unsigned nb = row[brick_nbr_code(dx, dy, dz)];
        if (nb >= a.n_bricks_cap) {
            int bx, by, bz;
            unpack_key(a.brick_key[bi], bx, by, bz);
            nb = brick_find(a, bx + dx, by + dy, bz + dz);
        }
        if (nb < a.n_bricks_cap)  // (exists by the invariant: ivox_upd_seq / the host build created it with the voxel)
            a.cells[nb * kBrickStride + brick_slab_index(sx - kBrickSide * dx, sy - kBrickSide * dy, sz - kBrickSide * dz)] = val;
    });
}
struct IvoxUpdBatch {
    const unsigned char* code;  // [n] 0 drop, 1 points_to_add, 2 point_no_need_downsample (ivox_add_decide_kernel)
    const float4* pw;           // [n] world points
    int n;
    uint2* lx;                  // [n] block-local exclusive counts {code 1, code 2}
    uint2* bt;                  // [blocks] block totals, then block offsets
    unsigned* seq_src;          // [A] source index of sequence rank r
    unsigned* seq_cell;         // [A] window cell of rank r (kUpdInvalidCell: outside)
    unsigned* jj;               // [A] arrival number of rank r inside its cell
    uint4* px;                  // [A] block-local exclusive {alloc, creations, touched, relocated capacity} of first-touchers
    uint4* bt2;                 // [blocks]
    unsigned char* fbit;        // [A] 1: rank r is the first point of this batch in its cell
    unsigned* tlist;            // [T] touched cells in first-touch order
};

__device__ __forceinline__ unsigned upd_cap_for(const unsigned n) {  // GridImage::cap_for
    unsigned c = 4;
    while (c < n) c <<= 1;
    return c;
}
__device__ __forceinline__ unsigned upd_log2(unsigned c) { return 31u - (unsigned)__clz((int)c); }

__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_count(const IvoxUpdBatch b) {
    __shared__ unsigned wsum[kUpdBlock / 64][2];
    const int i = blockIdx.x * kUpdBlock + threadIdx.x;
    const unsigned c = i < b.n ? b.code[i] : 0u;
    unsigned v[2] = {c == 1u ? 1u : 0u, c == 2u ? 1u : 0u}, tot[2];
    block_excl_scan<2>(v, tot, wsum);
    if (i < b.n) b.lx[i] = make_uint2(v[0], v[1]);
    if (threadIdx.x == 0) b.bt[blockIdx.x] = make_uint2(tot[0], tot[1]);
}

// one workgroup: exclusive scan of the block totals in place; batch totals into the state
__global__ void __launch_bounds__(kUpdMaxBlocks)
ivox_upd_scan1(const IvoxUpdBatch b, const int nblocks, IvoxUpdState* __restrict__ st) {
    __shared__ unsigned wsum[kUpdMaxBlocks / 64][2];
    const uint2 t = (int)threadIdx.x < nblocks ? b.bt[threadIdx.x] : make_uint2(0u, 0u);
    unsigned v[2] = {t.x, t.y}, tot[2];
    block_excl_scan<2>(v, tot, wsum);
    if ((int)threadIdx.x < nblocks) b.bt[threadIdx.x] = make_uint2(v[0], v[1]);
    if (threadIdx.x == 0) { st->n1 = tot[0]; st->n2 = tot[1]; st->status = kUpdOk; st->apply = 0u; }
}

// loads of words other threads of the SAME launch may have changed with atomics (the fused one-workgroup form below runs all phases
// in one launch: an L2 atomic does not update a line the CU's L1 already holds)
// COH = true only there (an agent-scope load is served from memory, ~3x the latency of an L2 hit: the multi-launch forms, whose phases
// are separated by kernel boundaries, use plain loads)
template <bool COH>
__device__ __forceinline__ unsigned upd_ld(const unsigned* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH>
__device__ __forceinline__ uint2 upd_ld_cell(const uint2* p) {
    if (!COH) return *p;
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)(v & 0xffffffffull), (unsigned)(v >> 32));
}

// sequence rank r = source point i: its cell (bricks created on demand), the sequence arrays, the per-cell scratch (count, first rank)
__device__ __forceinline__ void upd_seq_rank(const IvoxUpdBatch& b, const IvoxUpdArrays& a, IvoxUpdState* __restrict__ st, const unsigned r, const unsigned i) {
    const float4 p = b.pw[i];
    // IVoxMap::Pos2Grid (ivox_map.cpp:145-147): round half away from zero of the float product
    const float fx = roundf(p.x * a.inv_res), fy = roundf(p.y * a.inv_res), fz = roundf(p.z * a.inv_res);
    unsigned cell = kUpdInvalidCell;
    if (fabsf(fx) < (float)kKeyLimit && fabsf(fy) < (float)kKeyLimit && fabsf(fz) < (float)kKeyLimit) {
        // the voxel's brick, and -- for a voxel on its brick's boundary -- the neighbouring bricks whose halo mirrors it: all of them
        // exist from here on (created empty if need be)
        const int kx = (int)fx, ky = (int)fy, kz = (int)fz;
        const int bx = kx >> kBrickLog, by = ky >> kBrickLog, bz = kz >> kBrickLog;
        const int lx = kx & (kBrickSide - 1), ly = ky & (kBrickSide - 1), lz = kz & (kBrickSide - 1);
        const unsigned bi = brick_find_or_create(a, st, bx, by, bz);
        bool all = bi < a.n_bricks_cap;
        const int ex = lx == 0 ? -1 : lx == kBrickSide - 1 ? 1 : 0, ey = ly == 0 ? -1 : ly == kBrickSide - 1 ? 1 : 0, ez = lz == 0 ? -1 : lz == kBrickSide - 1 ? 1 : 0;
        if (all && (ex | ey | ez)) {
            // cached neighbour indices first (six independent loads in flight), the directory only for the ones never looked up
            unsigned* const row = a.nbr + (size_t)bi * 32u;
            unsigned v[6];
#pragma unroll
            for (int m = 1; m < 7; ++m) {
                const bool need = !(((m & 1) && !ex) || ((m & 2) && !ey) || ((m & 4) && !ez));
                v[m - 1] = need ? row[brick_nbr_code((m & 1) ? ex : 0, (m & 2) ? ey : 0, (m & 4) ? ez : 0)] : 0u;
            }
#pragma unroll
            for (int m = 1; m < 7; ++m) {
                const bool need = !(((m & 1) && !ex) || ((m & 2) && !ey) || ((m & 4) && !ez));
                if (!need) continue;
                unsigned nb = v[m - 1];
                if (nb >= a.n_bricks_cap) {
                    const int dx = (m & 1) ? ex : 0, dy = (m & 2) ? ey : 0, dz = (m & 4) ? ez : 0;
                    nb = brick_find_or_create(a, st, bx + dx, by + dy, bz + dz);
                    if (nb < a.n_bricks_cap) row[brick_nbr_code(dx, dy, dz)] = nb;
                }
                if (nb >= a.n_bricks_cap) all = false;
            }
        }
        if (all) cell = bi * kBrickStride + brick_slab_index(lx + 1, ly + 1, lz + 1);
    } else {
        atomicOr(&st->status, kUpdOutside);  // (a key beyond +-2^20: the host path reports FLS_ERR_RANGE)
    }
    b.seq_src[r] = i;
    b.seq_cell[r] = cell;
    if (cell == kUpdInvalidCell) { b.jj[r] = 0u; return; }  // (status already says why: kUpdOutside or kUpdArrayFull)
    b.jj[r] = atomicAdd(&a.pend[cell], 1u);
    atomicMin(&a.rank_mm[cell], r);
}
// top bit of a cell's `pend` word between ivox_evict_select and ivox_evict_apply (long chain only): the batch evicts this voxel BEFORE its
// first point arrives, i.e. the reference re-creates it from the batch's points alone (ivox_map.cpp:126-136) -- the second plan pass
// counts it as a creation with nothing to keep
constexpr unsigned kUpdRecreate = 0x80000000u;
// what the batch does to the voxel whose first point has rank r: {new region slots, creation, touched, capacity left behind}
template <bool COH = false>
__device__ __forceinline__ bool upd_plan_rank(const IvoxUpdBatch& b, const IvoxUpdArrays& a, const unsigned r, unsigned (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0u;
    const unsigned cell = b.seq_cell[r];
    if (cell == kUpdInvalidCell || upd_ld<COH>(&a.rank_mm[cell]) != r) return false;
    uint2 old = upd_ld_cell<COH>(&a.cells[cell]);
    const unsigned pe = upd_ld<COH>(&a.pend[cell]);
    const bool recreate = (pe & kUpdRecreate) != 0u;
    if (recreate) old = make_uint2(0u, 0u);  // (its old points and region are accounted for by the eviction)
    const unsigned total = old.y + (pe & ~kUpdRecreate);
    const unsigned cl = recreate ? 0u : a.cap_log2[cell];
    const unsigned cap = cl ? (1u << cl) : 0u;
    const bool grow = total > cap;
    v[0] = grow ? upd_cap_for(total) : 0u;
    v[1] = old.y == 0u ? 1u : 0u;
    v[2] = 1u;
    v[3] = grow ? cap : 0u;
    return true;
}
// the all-or-nothing verdict of a batch from its totals {alloc, creations, touched, relocated}: the status word and the number of evictions
// (a pure function of words no thread changes while it is evaluated: several workgroups may evaluate it side by side)
__device__ __forceinline__ unsigned upd_verdict(const IvoxUpdState* __restrict__ st, const unsigned (&tot)[4], const unsigned evict_ready, const unsigned n_list,
                                                unsigned status /* the batch's status word as the phases before left it */, unsigned& e) {
    // all-or-nothing: room in the point array.  LRU evictions inside the batch (ivox_map.cpp:133-136 evicts the list's back when
    // the count REACHES the capacity after a creation): with n alive voxels and k creations the batch evicts
    // E = max(0, n + k - (capacity - 1)) voxels; they are the E least recently touched ones as long as none of those is touched by
    // this batch (ivox_evict_check) -- the host queues the selection (alive cells sorted by stamp) whenever the batch could get there.
    if (st->used + (unsigned long long)tot[0] > st->pts_capacity) status |= kUpdArrayFull;
    const unsigned long long total = (unsigned long long)st->n_alive + tot[1];
    e = 0u;
    if (total >= (unsigned long long)st->lru_capacity) {
        e = (unsigned)(total - (unsigned long long)st->lru_capacity + 1ull);
        if (!evict_ready || e > n_list) status |= kUpdNeedHost;
    }
    return status;
}
__device__ __forceinline__ void upd_decide_totals(IvoxUpdState* __restrict__ st, const unsigned (&tot)[4], const unsigned evict_ready, const unsigned n_list,
                                                  const unsigned status_in) {
    unsigned e;
    const unsigned status = upd_verdict(st, tot, evict_ready, n_list, status_in, e);
    st->alloc = tot[0]; st->creations = tot[1]; st->touched = tot[2]; st->relocated_garbage = tot[3];
    st->evict_ready = evict_ready;  // (round 3: a one-thread launch of their own used to set these two words)
    st->n_list = n_list;
    st->evict = e;
    st->evicted_points = 0ull;
    st->evicted_slots = 0ull;
    st->recreated = 0u;
    st->status = status;
    if (!evict_ready) st->apply = status == kUpdOk ? 1u : 0u;  // no eviction selection follows: this is the verdict (ivox_upd_decide otherwise)
}
// the slot region of the voxel first touched by rank r: relocated when it outgrows its capacity; cell + halo copies; touched list
template <bool COH = false>
__device__ __forceinline__ void upd_region_rank(const IvoxUpdBatch& b, const IvoxUpdArrays& a, const IvoxUpdState* __restrict__ st, const unsigned r,
                                                const unsigned alloc_before, const unsigned touched_before) {
    const unsigned cell = b.seq_cell[r];
    const uint2 old = upd_ld_cell<COH>(&a.cells[cell]);
    const unsigned total = old.y + upd_ld<COH>(&a.pend[cell]);
    const unsigned cl = a.cap_log2[cell];
    const unsigned cap = cl ? (1u << cl) : 0u;
    unsigned begin = old.x;
    if (total > cap) {  // grown past its region (or new): a fresh region at the end of the array, in first-touch order
        const unsigned ncap = upd_cap_for(total);
        begin = (unsigned)st->used + alloc_before;
        for (unsigned k = 0; k < old.y; ++k) a.pts[begin + k] = a.pts[old.x + k];
        for (unsigned k = total; k < ncap; ++k) a.pts[begin + k] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));  // slack
        a.cap_log2[cell] = (unsigned char)upd_log2(ncap);
    }
    a.cells[cell] = make_uint2(begin, total);
    brick_write_mirrors(a, cell, make_uint2(begin, total));
    b.tlist[touched_before] = cell;
}
template <bool COH = false>
__device__ __forceinline__ void upd_point_rank(const IvoxUpdBatch& b, const IvoxUpdArrays& a, const IvoxUpdState* __restrict__ st, const unsigned r) {
    const unsigned cell = b.seq_cell[r];
    const uint2 e = upd_ld_cell<COH>(&a.cells[cell]);
    const float4 p = b.pw[b.seq_src[r]];
    a.pts[e.x + (e.y - upd_ld<COH>(&a.pend[cell])) + b.jj[r]] = make_float4(p.x, p.y, p.z, __int_as_float(st->next_id + (int)r));
}

// rank of every inserted point, its window cell, the sequence arrays, and the per-cell scratch (count, first rank)
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_seq(const IvoxUpdBatch b, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st) {
    const int i = blockIdx.x * kUpdBlock + threadIdx.x;
    if (i >= b.n) return;
    const unsigned c = b.code[i];
    if (c == 0u) return;
    const uint2 l = b.lx[i], base = b.bt[blockIdx.x];
    const unsigned r = c == 1u ? base.x + l.x : st->n1 + base.y + l.y;
    upd_seq_rank(b, a, st, r, (unsigned)i);
}

// first-toucher flags and, per touched voxel, what the batch does to it; block-local scans of the four counters
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_plan(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st) {
    __shared__ unsigned wsum[kUpdBlock / 64][4];
    if (st->status & kUpdSkipped) return;
    const unsigned A = st->n1 + st->n2;
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    unsigned v[4] = {0u, 0u, 0u, 0u}, tot[4];
    if (r < A) b.fbit[r] = upd_plan_rank(b, a, r, v) ? 1 : 0;
    block_excl_scan<4>(v, tot, wsum);
    if (r < A) b.px[r] = make_uint4(v[0], v[1], v[2], v[3]);
    if (threadIdx.x == 0) b.bt2[blockIdx.x] = make_uint4(tot[0], tot[1], tot[2], tot[3]);
}

__global__ void __launch_bounds__(kUpdMaxBlocks)
ivox_upd_scan2(const IvoxUpdBatch b, IvoxUpdState* __restrict__ st, const unsigned evict_ready /* the host queued the eviction selection */,
               const unsigned n_list /* entries of the stamp-sorted list of alive cells */) {
    __shared__ unsigned wsum[kUpdMaxBlocks / 64][4];
    const unsigned A = st->n1 + st->n2;
    const int nblocks = (int)((A + kUpdBlock - 1) / kUpdBlock);
    const uint4 t = (int)threadIdx.x < nblocks ? b.bt2[threadIdx.x] : make_uint4(0u, 0u, 0u, 0u);
    unsigned v[4] = {t.x, t.y, t.z, t.w}, tot[4];
    block_excl_scan<4>(v, tot, wsum);
    if ((int)threadIdx.x < nblocks) b.bt2[threadIdx.x] = make_uint4(v[0], v[1], v[2], v[3]);
    if (threadIdx.x == 0) upd_decide_totals(st, tot, evict_ready, n_list, st->status);
}

// ---- eviction selection: the alive cells of the image as a list sorted by LRU stamp ------------------------------------------
constexpr int kEvBlock = 1024;
__device__ __forceinline__ bool brick_cell_is_primary(const unsigned c) {  // an interior cell of its slab (halo cells are mirrors)
    int sx, sy, sz;
    return brick_slab_interior(c & (kBrickStride - 1u), sx, sy, sz);
}
// pass 1: alive cells per block of kEvBlock cells
__global__ void __launch_bounds__(kEvBlock)
ivox_evict_count(const uint2* __restrict__ cells, const unsigned ncell, unsigned* __restrict__ bt) {
    __shared__ unsigned wsum[kEvBlock / 64];
    const unsigned c = blockIdx.x * kEvBlock + threadIdx.x;
    const bool alive = c < ncell && brick_cell_is_primary(c) && cells[c].y != 0u;
    const unsigned long long m = __ballot(alive);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { unsigned t = 0; for (int q = 0; q < kEvBlock / 64; ++q) t += wsum[q]; bt[blockIdx.x] = t; }
}
// pass 2: {low word of the stamp, cell} of every alive cell, in cell order (bt holds the scanned block offsets)
__global__ void __launch_bounds__(kEvBlock)
ivox_evict_list(const uint2* __restrict__ cells, const unsigned long long* __restrict__ stamp, const unsigned ncell, const unsigned* __restrict__ bt,
                unsigned* __restrict__ key, unsigned* __restrict__ val) {
    __shared__ unsigned wsum[kEvBlock / 64];
    const unsigned c = blockIdx.x * kEvBlock + threadIdx.x;
    const bool alive = c < ncell && brick_cell_is_primary(c) && cells[c].y != 0u;
    const unsigned long long m = __ballot(alive);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned base = bt[blockIdx.x];
    for (int q = 0; q < w; ++q) base += wsum[q];
    if (alive) {
        const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        key[pos] = (unsigned)stamp[c];
        val[pos] = c;
    }
}
// second sort round: the high word of the stamps, read through the order of the first round
__global__ void __launch_bounds__(256)
ivox_evict_hikeys(const unsigned long long* __restrict__ stamp, const unsigned* __restrict__ order, const unsigned n, unsigned* __restrict__ key) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) key[i] = (unsigned)(stamp[order[i]] >> 32);
}
// rank of every creation of the batch, in creation order: the i-th eviction happens right after creation number
// (capacity - 1 - n_alive) + i (ivox_map.cpp:133-136)
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_cranks(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st, unsigned* __restrict__ crank) {
    const unsigned A = st->n1 + st->n2;
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    if (r >= A || !b.fbit[r]) return;
    const unsigned cell = b.seq_cell[r];
    if (a.cells[cell].y != 0u) return;  // an existing voxel: no creation
    crank[b.bt2[blockIdx.x].y + b.px[r].y] = r;
}
// Which voxels the batch evicts.  The candidates are the alive cells in LRU order (oldest first); eviction number idx happens right
// after creation number base_c + idx of the batch (ivox_map.cpp:133-136: the creation that brings the count to the capacity evicts
// the list's back).  An UNTOUCHED candidate is the next eviction.  A candidate the batch touches has moved to the list's front by
// the time the eviction pointer reaches it iff its first touch (rank_mm = first rank, after ivox_upd_seq) precedes that eviction's
// creation: the reference skips it, and so does this walk.  If the eviction comes first the reference evicts the voxel and the later
// touch RE-CREATES it from the batch's points alone: one more creation -- its rank joins the creation sequence -- and therefore one
// more eviction.  Round 4 resolves that here (it used to refuse the batch): the candidate is evicted, marked kUpdRecreate for the
// second plan pass, its first rank is inserted into the sorted list S, and the walk continues with creation ranks read from the
// MERGED sequence crank U S.  A marked voxel's turn is found one at a time (everything decided in parallel behind an undiscovered one
// would use the wrong sequence): the chunk is re-evaluated from the position behind it -- conflicts are rare, a pass is 1024 candidates.
// The walk is checked against the sequential loop on random maps in tests/host/evict_conflict_model_test.cpp.  What still goes to the
// host: a selection that runs out of untouched-or-late candidates (it would reach voxels this batch itself touched or created:
// kUpdNeedHost) and more than kEvMaxRecreate re-created voxels in one batch.  One workgroup.
constexpr int kEvMaxRecreate = 1024;
__global__ void __launch_bounds__(kEvBlock)
ivox_evict_select(const unsigned* __restrict__ order, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st, const unsigned* __restrict__ crank,
                  unsigned* __restrict__ evict_list) {
    __shared__ unsigned wsum[kEvBlock / 64];
    __shared__ unsigned s_found, s_nS, s_conflict, s_overflow;
    __shared__ unsigned s_S[kEvMaxRecreate];
    const unsigned E = st->evict;
    if (E == 0u || st->status != kUpdOk) return;
    const unsigned n_list = st->n_list, C = st->creations;
    const unsigned base_c = st->lru_capacity - 1u > st->n_alive ? st->lru_capacity - 1u - st->n_alive : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_found = 0u; s_nS = 0u; s_conflict = 0xFFFFFFFFu; s_overflow = 0u; }
    __syncthreads();
    for (unsigned j0 = 0; j0 < n_list; j0 += kEvBlock) {
        if (s_found >= E + s_nS) break;
        const unsigned j = j0 + threadIdx.x;
        const bool valid = j < n_list;
        const unsigned cell = valid ? order[j] : 0u;
        const bool un = valid && a.pend[cell] == 0u;
        const unsigned rv = (valid && !un) ? a.rank_mm[cell] : 0u;  // first rank of a touched candidate
        unsigned start = j0;  // positions of this chunk in front of `start` are decided
        for (;;) {
            const unsigned found = s_found, nS = s_nS, Etot = E + nS;  // (uniform: written before the last barrier)
            if (found >= Etot) break;
            const bool in = valid && j >= start;
            const unsigned long long m = __ballot(in && un);
            if (lane == 0) wsum[w] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned before = 0u, total = 0u;
            for (int q = 0; q < kEvBlock / 64; ++q) { const unsigned t = wsum[q]; if (q < w) before += t; total += t; }
            const unsigned idx = found + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));  // evictions decided before this candidate
            // evicted first, re-created later?  (only the FIRST such position of the pass is trusted)
            if (in && !un && idx < Etot && !(rv < evict_merged_rank(crank, C, s_S, nS, base_c + idx))) atomicMin(&s_conflict, j);
            __syncthreads();
            const unsigned fc = s_conflict;
            if (in && un && j < fc && idx < Etot) evict_list[idx] = cell;
            __syncthreads();  // (every thread has read s_conflict / s_found / s_nS / wsum)
            if (fc == 0xFFFFFFFFu) {
                if (threadIdx.x == 0) s_found = found + total;
                __syncthreads();
                break;
            }
            if (j == fc) {  // this candidate: evicted as number idx, re-created by its first point
                evict_list[idx] = cell;
                if (nS >= (unsigned)kEvMaxRecreate) {
                    s_overflow = 1u;
                } else {
                    a.pend[cell] |= kUpdRecreate;
                    unsigned q = nS;
                    while (q > 0u && s_S[q - 1] > rv) { s_S[q] = s_S[q - 1]; --q; }
                    s_S[q] = rv;
                    s_nS = nS + 1u;
                }
                s_found = idx + 1u;
                s_conflict = 0xFFFFFFFFu;
            }
            __syncthreads();
            if (s_overflow) break;
            start = fc + 1u;
        }
        if (s_overflow) break;
    }
    if (threadIdx.x == 0) {
        if (s_overflow) atomicOr(&st->status, kUpdEvictConflict);  // (more re-created voxels than the list holds: the sequential host code)
        else if (s_found < E + s_nS) atomicOr(&st->status, kUpdNeedHost);
        else { st->evict = E + s_nS; st->recreated = s_nS; }
    }
}
// second plan pass of a batch with an eviction selection (after ivox_upd_plan has run again, now seeing the kUpdRecreate marks): the
// block offsets and totals as the regions will be laid out, the point-array check with them, and the cross-check that plan and selection
// agree on the number of evictions.  Without marked voxels everything comes out as in the first pass.
__global__ void __launch_bounds__(kUpdMaxBlocks)
ivox_upd_scan2_again(const IvoxUpdBatch b, IvoxUpdState* __restrict__ st) {
    __shared__ unsigned wsum[kUpdMaxBlocks / 64][4];
    const unsigned A = st->n1 + st->n2;
    const int nblocks = (int)((A + kUpdBlock - 1) / kUpdBlock);
    const uint4 t = (int)threadIdx.x < nblocks ? b.bt2[threadIdx.x] : make_uint4(0u, 0u, 0u, 0u);
    unsigned v[4] = {t.x, t.y, t.z, t.w}, tot[4];
    block_excl_scan<4>(v, tot, wsum);
    if ((int)threadIdx.x < nblocks) b.bt2[threadIdx.x] = make_uint4(v[0], v[1], v[2], v[3]);
    if (threadIdx.x == 0) {
        unsigned e;
        unsigned status = upd_verdict(st, tot, st->evict_ready, st->n_list, st->status, e);
        if (status == kUpdOk && e != st->evict) status |= kUpdNeedHost;  // (never expected: the selection counted E + re-created)
        st->alloc = tot[0]; st->creations = tot[1]; st->touched = tot[2]; st->relocated_garbage = tot[3];
        st->status = status;
    }
}
__global__ void ivox_upd_decide(IvoxUpdState* __restrict__ st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st->apply = st->status == kUpdOk ? 1u : 0u;
}
__global__ void __launch_bounds__(256)
ivox_evict_apply(const unsigned* __restrict__ evict_list, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st) {
    const unsigned j = blockIdx.x * 256 + threadIdx.x;
    if (!st->apply || j >= st->evict) return;
    const unsigned cell = evict_list[j];
    const uint2 e = a.cells[cell];
    const unsigned cl = a.cap_log2[cell];
    atomicAdd(&st->evicted_points, (unsigned long long)e.y);
    atomicAdd(&st->evicted_slots, cl ? (unsigned long long)(1u << cl) : 0ull);
    a.cells[cell] = make_uint2(0u, 0u);  // the region's slots are garbage from here on (never read: count 0)
    brick_write_mirrors(a, cell, make_uint2(0u, 0u));
    a.cap_log2[cell] = 0;
    a.stamp[cell] = 0ull;
    a.pend[cell] &= ~kUpdRecreate;  // (a re-created voxel is an empty cell with pending points from here on: regions / points / finish treat it as new)
}

// the scratch word of every touched cell turns from the FIRST into the LAST rank of the batch (max >= min: one atomicMax);
// a batch that is not applied only resets its scratch
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_last(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st) {
    const unsigned A = st->n1 + st->n2;
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    if (r >= A) return;
    const unsigned cell = b.seq_cell[r];
    if (cell == kUpdInvalidCell) return;
    if (st->apply) atomicMax(&a.rank_mm[cell], r);
    else { a.pend[cell] = 0u; a.rank_mm[cell] = kUpdNoRank; }  // (the same values from every point of the cell)
}

__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_regions(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st) {
    if (!st->apply) return;
    const unsigned A = st->n1 + st->n2;
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    if (r >= A || !b.fbit[r]) return;
    const uint4 loc = b.px[r], base = b.bt2[blockIdx.x];
    upd_region_rank(b, a, st, r, base.x + loc.x, base.z + loc.z);
}

__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_points(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st) {
    if (!st->apply) return;
    const unsigned A = st->n1 + st->n2;
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    if (r >= A) return;
    upd_point_rank(b, a, st, r);
}

// per touched voxel: its new points into insertion (id) order, LRU stamp, scratch reset.  ONE WAVE per voxel: the new points are
// staged in LDS, every lane counts for its points how many of the voxel's new ids are smaller (= the point's final position) and
// writes them back in place.  (A voxel next to the sensor receives hundreds of points from one scan: the first version, one thread
// per voxel with an insertion sort in global memory, took up to 1.2 ms.)
constexpr int kUpdFinishMaxK = 1024;  // new points of one voxel staged per wave; more than that: serial fallback by lane 0
// one wave: the new points of touched voxel number t into id order (s_row: MAXK float4 of LDS owned by this wave), stamp, scratch reset
template <int MAXK, bool COH = false>
__device__ __forceinline__ void upd_finish_voxel(const IvoxUpdBatch& b, const IvoxUpdArrays& a, const IvoxUpdState* __restrict__ st, const unsigned t,
                                                 float4* __restrict__ s_row, const int lane) {
    const unsigned cell = b.tlist[t];
    const uint2 e = upd_ld_cell<COH>(&a.cells[cell]);
    const unsigned k = upd_ld<COH>(&a.pend[cell]);
    float4* const q = a.pts + e.x + (e.y - k);
    if (k > 1u && k <= (unsigned)MAXK) {
        for (unsigned i = lane; i < k; i += 64) s_row[i] = q[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (unsigned i = lane; i < k; i += 64) {
            const float4 x = s_row[i];
            const int id = __float_as_int(x.w);
            unsigned pos = 0;
            for (unsigned j = 0; j < k; ++j) pos += __float_as_int(s_row[j].w) < id ? 1u : 0u;  // ids are distinct
            q[pos] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the row is reused by the wave's next voxel in the fused form)
    } else if (k > (unsigned)MAXK && lane == 0) {
        for (unsigned i = 1; i < k; ++i) {  // (rare: more new points in one 0.5 m voxel than the wave stages)
            const float4 x = q[i];
            const int id = __float_as_int(x.w);
            unsigned j = i;
            while (j > 0 && __float_as_int(q[j - 1].w) > id) { q[j] = q[j - 1]; --j; }
            q[j] = x;
        }
    }
    if (lane == 0) {
        a.stamp[cell] = st->stamp_base + upd_ld<COH>(&a.rank_mm[cell]) + 1ull;
        a.pend[cell] = 0u;
        a.rank_mm[cell] = kUpdNoRank;
    }
}
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_finish(const IvoxUpdBatch b, const IvoxUpdArrays a, const IvoxUpdState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ float4 s_pts[kUpdBlock / 64][kUpdFinishMaxK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned t = blockIdx.x * (kUpdBlock / 64) + w;  // one wave per touched voxel
    if (t >= st->touched) return;
    upd_finish_voxel<kUpdFinishMaxK>(b, a, st, t, &s_pts[w][0], lane);
}

__device__ __forceinline__ void upd_commit(IvoxUpdState* __restrict__ st, IvoxUpdMailbox* __restrict__ mb, const unsigned seq, const unsigned n_bricks_cap) {
    const unsigned A = (st->status & kUpdSkipped) ? 0u : st->n1 + st->n2;
    if (st->n_bricks > n_bricks_cap) st->n_bricks = n_bricks_cap;  // (a refused batch overshot the pool)
    if (st->apply) {
        st->n_alive += st->creations;
        st->n_alive -= st->evict;
        st->n_points += A;
        st->n_points -= st->evicted_points;
        st->next_id += (int)A;
        st->used += st->alloc;
        st->garbage += st->relocated_garbage + st->evicted_slots;
        st->stamp_base += A;
    }
    __hip_atomic_store(&mb->n_bricks, st->n_bricks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->n_points, st->n_points, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->used, st->used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->garbage, st->garbage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->n_alive, st->n_alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->status, st->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->added, st->apply ? A : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->touched, st->touched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->next_id, st->next_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->evicted, st->apply ? st->evict : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->recreated, st->apply ? st->recreated : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void ivox_upd_commit(IvoxUpdState* __restrict__ st, IvoxUpdMailbox* __restrict__ mb, const unsigned seq, const unsigned n_bricks_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    upd_commit(st, mb, seq, n_bricks_cap);
}

// ---- the short chain (round 4): seven dependent launches instead of eleven --------------------------------------------------------
// A dependent launch costs ~7 us on this path (2-5 us of work + ~4 us of dispatch latency), so the launches that only scanned a
// few hundred block totals, flipped first -> last ranks or published the result are folded into their neighbours:
//   ivox_add_decide_kernel (+ count)  ->  ivox_upd_seq_nb (every block sums the block totals before it: no scan1)  ->  ivox_upd_plan
//   ->  ivox_upd_last_regions (every block sums the plan totals and evaluates the verdict itself: no scan2, no separate `last`)
//   ->  ivox_upd_points  ->  ivox_upd_finish  ->  ivox_upd_commit   (ivox_upd_finish_commit, where the last block to finish publishes, is
//   the FLS_IVOX_FUSED_COMMIT=1 variant: measured slower, 15-26 us against 4.6 + 4.1 us).
// Same device functions, same arithmetic.  Batches that may evict keep the long chain (the selection needs grid-wide steps of its own).
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_seq_nb(const IvoxUpdBatch b, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st, const int nblocks) {
    __shared__ unsigned wsum[kUpdBlock / 64][4];
    if (st->status & kUpdSkipped) return;
    unsigned pre[2], all[2];
    block_prefix_total<2>(b.bt, nblocks, (int)blockIdx.x, pre, all, wsum);  // bt: raw block totals of the decision codes (ivox_add_decide_kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->n1 = all[0]; st->n2 = all[1]; }
    const int i = blockIdx.x * kUpdBlock + threadIdx.x;
    if (i >= b.n) return;
    const unsigned c = b.code[i];
    if (c == 0u) return;
    const uint2 l = b.lx[i];
    const unsigned r = c == 1u ? pre[0] + l.x : all[0] + pre[1] + l.y;
    upd_seq_rank(b, a, st, r, (unsigned)i);
}
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_last_regions(const IvoxUpdBatch b, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st) {
    __shared__ unsigned wsum[kUpdBlock / 64][8];
    if (st->status & kUpdSkipped) return;
    const unsigned A = st->n1 + st->n2;
    const int nblocks = (int)((A + kUpdBlock - 1) / kUpdBlock);
    if ((int)blockIdx.x >= nblocks && blockIdx.x != 0) return;  // (block 0 always runs: it records the verdict, also of an empty batch)
    unsigned pre[4], tot[4], e;
    block_prefix_total<4>(b.bt2, nblocks, (int)blockIdx.x, pre, tot, wsum);  // bt2: raw block totals of ivox_upd_plan
    const unsigned status_in = st->status;  // (left by ivox_upd_seq_nb; block 0 stores the same bits back below)
    const bool apply = upd_verdict(st, tot, 0u, 0u, status_in, e) == kUpdOk;  // the same words in every block: the same verdict
    const unsigned r = blockIdx.x * kUpdBlock + threadIdx.x;
    if (r < A) {
        const unsigned cell = b.seq_cell[r];
        if (cell != kUpdInvalidCell) {
            if (apply) atomicMax(&a.rank_mm[cell], r);
            else { a.pend[cell] = 0u; a.rank_mm[cell] = kUpdNoRank; }
        }
        if (apply && b.fbit[r]) { const uint4 loc = b.px[r]; upd_region_rank(b, a, st, r, pre[0] + loc.x, pre[2] + loc.z); }
    }
    // the record of the verdict (read by the launches that follow); written last: the blocks above only READ the words it is made of,
    // and the status it stores is the one they computed
    if (blockIdx.x == 0 && threadIdx.x == 0) upd_decide_totals(st, tot, 0u, 0u, status_in);
}
// points of rank r in the short chain: the block offsets of ivox_upd_plan were never scanned in place, nothing else differs
// (ivox_upd_points serves both chains)
constexpr int kFinishBlocks = 256;
__global__ void __launch_bounds__(kUpdBlock)
ivox_upd_finish_commit(const IvoxUpdBatch b, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st, IvoxUpdMailbox* __restrict__ mb, const unsigned seq,
                       unsigned* __restrict__ ticket) {
    __shared__ float4 s_pts[kUpdBlock / 64][kUpdFinishMaxK];
    __shared__ unsigned s_last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (st->apply) {
        const unsigned touched = st->touched;
        for (unsigned t = blockIdx.x * (kUpdBlock / 64) + w; t < touched; t += gridDim.x * (kUpdBlock / 64)) upd_finish_voxel<kUpdFinishMaxK>(b, a, st, t, &s_pts[w][0], lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = fanin_last_arriver(ticket, 8);
    __syncthreads();
    if (s_last && threadIdx.x == 0) upd_commit(st, mb, seq, a.n_bricks_cap);  // every other block is done with the words this changes
}

// ---- the whole batch in ONE launch of ONE workgroup (small batches: the 0.5 m-filtered planar cloud the pipeline feeds, ~10 k points) ----
// The ten launches above are 2-5 us of work each with ~4 us of dispatch latency between dependent launches: ~90 us for microseconds of
// work.  Up to kFusedMaxN source points the same phases run inside one 1024-thread workgroup, separated by
// __syncthreads() instead of kernel boundaries: the two block scans happen once (16 consecutive codes per thread; a contiguous slice of
// the ranks per thread), the per-rank phases stride over the ranks.  Same device functions, same arithmetic, same result -- ranks,
// regions in first-touch order, ids, stamps.  Batches that may evict (the selection sorts the alive cells: a grid-wide job) and larger
// batches keep the multi-launch form.
constexpr int kFusedThreads = 1024, kFusedItems = 16, kFusedMaxN = kFusedThreads * kFusedItems, kFusedFinishK = 256;
__global__ void __launch_bounds__(kFusedThreads)
ivox_upd_fused_kernel(const IvoxUpdBatch b, const IvoxUpdArrays a, IvoxUpdState* __restrict__ st, IvoxUpdMailbox* __restrict__ mb, const unsigned seq) {
    __shared__ unsigned wsum2[kFusedThreads / 64][2];
    __shared__ unsigned wsum4[kFusedThreads / 64][4];
    __shared__ uint4 s_base[kFusedThreads];
    __shared__ float4 s_pts[kFusedThreads / 64][kFusedFinishK];
    __shared__ unsigned s_apply;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) { st->status = kUpdOk; st->apply = 0u; }
    // ---- count + rank: 16 consecutive decision codes per thread ----
    unsigned char cd[kFusedItems];
    unsigned v2[2] = {0u, 0u}, tot2[2];
#pragma unroll
    for (int k = 0; k < kFusedItems; ++k) {
        const int i = t * kFusedItems + k;
        cd[k] = i < b.n ? b.code[i] : (unsigned char)0;
        v2[0] += cd[k] == 1 ? 1u : 0u;
        v2[1] += cd[k] == 2 ? 1u : 0u;
    }
    block_excl_scan<2>(v2, tot2, wsum2);  // (its barriers also order the status reset above before any atomicOr below)
    const unsigned n1 = tot2[0], A = tot2[0] + tot2[1];
    {
        unsigned r1 = v2[0], r2 = n1 + v2[1];
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            const unsigned i = (unsigned)(t * kFusedItems + k);
            if (cd[k] == 1) b.seq_src[r1++] = i;
            else if (cd[k] == 2) b.seq_src[r2++] = i;
        }
    }
    if (t == 0) { st->n1 = n1; st->n2 = tot2[1]; }
    __syncthreads();
    // ---- seq: cell of every rank (bricks on demand), arrival number, first rank per cell ----
    for (unsigned r = t; r < A; r += kFusedThreads) upd_seq_rank(b, a, st, r, b.seq_src[r]);
    __syncthreads();
    // ---- plan: a contiguous slice of the ranks per thread; exclusive {alloc, creations, touched, relocated} of the first-touchers ----
    const unsigned per = (A + kFusedThreads - 1) / kFusedThreads, r_lo = min(A, (unsigned)t * per), r_hi = min(A, r_lo + per);
    unsigned v4[4] = {0u, 0u, 0u, 0u}, tot4[4];
    for (unsigned r = r_lo; r < r_hi; ++r) {
        unsigned inc[4];
        const bool first = upd_plan_rank<true>(b, a, r, inc);
        b.fbit[r] = first ? 1 : 0;
        b.px[r] = make_uint4(v4[0], v4[1], v4[2], v4[3]);  // exclusive inside the slice
        v4[0] += inc[0]; v4[1] += inc[1]; v4[2] += inc[2]; v4[3] += inc[3];
    }
    block_excl_scan<4>(v4, tot4, wsum4);
    s_base[t] = make_uint4(v4[0], v4[1], v4[2], v4[3]);
    if (t == 0) {
        upd_decide_totals(st, tot4, 0u, 0u, __hip_atomic_load(&st->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));  // (no eviction selection in this form: a batch that reaches the capacity is sent to the multi-launch form by the host)
        s_apply = st->apply;
    }
    __syncthreads();
    const bool apply = s_apply != 0u;
    // ---- last: first rank -> last rank per cell, or scratch reset when the batch is refused ----
    for (unsigned r = t; r < A; r += kFusedThreads) {
        const unsigned cell = b.seq_cell[r];
        if (cell == kUpdInvalidCell) continue;
        if (apply) atomicMax(&a.rank_mm[cell], r);
        else { a.pend[cell] = 0u; a.rank_mm[cell] = kUpdNoRank; }
    }
    if (apply) {
        __syncthreads();
        // ---- regions (slice order = first-touch order), points, per-voxel id order ----
        const uint4 base = s_base[t];
        for (unsigned r = r_lo; r < r_hi; ++r)
            if (b.fbit[r]) { const uint4 loc = b.px[r]; upd_region_rank<true>(b, a, st, r, base.x + loc.x, base.z + loc.z); }
        __syncthreads();
        for (unsigned r = t; r < A; r += kFusedThreads) upd_point_rank<true>(b, a, st, r);
        __syncthreads();
        const unsigned touched = tot4[2];
        for (unsigned tv = w; tv < touched; tv += kFusedThreads / 64) upd_finish_voxel<kFusedFinishK, true>(b, a, st, tv, &s_pts[w][0], lane);
    }
    __syncthreads();
    if (t == 0) upd_commit(st, mb, seq, a.n_bricks_cap);
}

// ---- image <-> host mirror ---------------------------------------------------------------------------------------------------------
// every alive voxel of the image as a record (sync_host_from_device: the device image back into the host mirror), in any order
struct IvoxAliveRec { unsigned long long key; unsigned begin, count, cap_log2, pad; unsigned long long stamp; };
__global__ void __launch_bounds__(256)
ivox_list_alive_kernel(const uint2* __restrict__ cells, const unsigned long long* __restrict__ brick_key, const unsigned ncell,
                       const unsigned char* __restrict__ cap_log2, const unsigned long long* __restrict__ stamp, IvoxAliveRec* __restrict__ out,
                       unsigned* __restrict__ counter, const unsigned out_cap) {
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    int sx = 0, sy = 0, sz = 0;
    const bool prim = c < ncell && brick_slab_interior(c & (kBrickStride - 1u), sx, sy, sz);
    const uint2 e = prim ? cells[c] : make_uint2(0u, 0u);
    const bool alive = prim && e.y != 0u;
    const unsigned long long m = __ballot(alive);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0u;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (!alive) return;
    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (pos >= out_cap) return;
    int bx, by, bz;
    unpack_key(brick_key[c / kBrickStride], bx, by, bz);
    out[pos] = IvoxAliveRec{pack_key(bx * kBrickSide + sx - 1, by * kBrickSide + sy - 1, bz * kBrickSide + sz - 1), e.x, e.y, (unsigned)cap_log2[c], 0u, stamp[c]};
}
// per-voxel update metadata scattered into the (zeroed) per-cell arrays (enter_device_mode)
struct IvoxMetaRec { unsigned cell, cap_log2; unsigned long long stamp; };
__global__ void __launch_bounds__(256)
ivox_meta_scatter_kernel(const IvoxMetaRec* __restrict__ rec, const unsigned n, unsigned char* __restrict__ cap_log2, unsigned long long* __restrict__ stamp) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const IvoxMetaRec r = rec[i];
    cap_log2[r.cell] = (unsigned char)r.cap_log2;
    stamp[r.cell] = r.stamp;
}

}  // namespace fls
