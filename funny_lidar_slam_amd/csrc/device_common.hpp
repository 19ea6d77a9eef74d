// device_common.hpp -- device-side data layout shared by all kernels (gfx950).
//
// HBM layout (DESIGN.md "Data layout"):
//   * source scan        : SoA  sx[n], sy[n], sz[n]  (float)  -> lane i reads element i: coalesced
//   * voxel / cell table : open-addressing hash, 16 B entries {u64 packed key, u32 begin, u32 count}
//                          one dwordx4 load per probe; load factor <= 0.5
//   * map points         : float4 {x, y, z, id-as-int-bits}, bucketed by voxel (CSR order), one
//                          dwordx4 load per candidate (gather, so AoS-of-16B beats SoA here)
//   * Gauss-Newton state : one GnState block (pose, stop flags, per-iteration log), device resident
//                          for the whole Match; the host reads it back once at the end
//   * workgroup partials : 32 doubles per workgroup (21 upper-H + 6 g + res + count + 3 spare)
//   * result mailbox     : host-mapped pinned memory written by the last workgroup (Mailbox)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fls {

constexpr int kMaxIter = 64;
constexpr int kPartialStride = 32;  // doubles per wave partial row
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kKeyLimit = (1 << 20) - 2;  // |voxel key| bound per axis (21-bit packing)

struct HashEntry {
    unsigned long long key;
    unsigned begin;
    unsigned count;
};
static_assert(sizeof(HashEntry) == 16, "one dwordx4 per probe");

struct DevGrid {
    const HashEntry* table;
    const float4* pts;  // xyz + id bits
    unsigned mask;      // table size - 1 (power of two)
    unsigned n_pts;
};

// Dense voxel window (optional, iVox path): a plain 3-D array of {begin, count} over the bounding box of the
// occupied voxel keys, x fastest.  One aligned 8-byte load per probe, no hashing, and -- unlike the hash
// table, which scatters neighbouring voxels over the whole table -- neighbouring voxels share cache lines,
// so the per-XCD L2 (4 MiB) holds the slice of the map its queries touch.
struct DenseWindow {
    const uint2* cells;  // nullptr: window not built (extent too large) -> hash table path
    int ox, oy, oz;      // voxel key of cell (0,0,0)
    int nx, ny, nz;
};

// Two-level voxel image of the iVox map (round 4; replaces the bounding-box window for the iVox kind -- the reference's IVoxMap bounds the
// NUMBER of voxels, never the extent, include/ivox_map/ivox_map.h:35): the voxel key space is cut into BRICKS of 8 x 8 x 8 voxels; a brick
// that holds (or borders) an occupied voxel owns a slab of {begin, count} cells, found through a small open-addressing directory
// {packed brick key, brick index}.  A slab stores 10 x 10 x 10 cells: the brick's own 8^3 plus a one-cell HALO that mirrors the boundary
// cells of the neighbouring bricks, so that all 19 NEARBY18 probes of a query resolve inside the slab of the query's own brick -- ONE
// directory look-up per query (one aligned 16-byte load, the same address for the lanes of a group), then plain slab offsets, no bounds
// checks.  Invariant kept by every writer (host build, journal scatter, device AddPoints / evictions): a halo cell equals the interior
// cell it mirrors, and a brick exists whenever one of its interior or (face / edge) halo cells is occupied -- so "brick missing" means
// "no candidate in any of the 19 voxels".  NEARBY18 has no (+-1, +-1, +-1) offsets: corner halo cells are never read nor maintained.
// Extent independent: bricks anywhere in the +-2^20 key range; a 1e6-point map is ~3.5 k bricks = 28 MB of cells (the bounding-box
// window it replaces: 17 M cells = 136 MB) and a 56 KB directory.
constexpr int kBrickLog = 3;
constexpr int kBrickSide = 1 << kBrickLog;
constexpr int kBrickStored = kBrickSide + 2;   // interior + halo
constexpr unsigned kBrickStride = 1024u;       // cells per slab (1000 used), also the stride of every per-cell side array
constexpr unsigned kBrickPending = 0xFFFFFFFFu;  // directory entry claimed, index not yet published (device-side creation)
constexpr unsigned kBrickInvalid = 0xFFFFFFFEu;  // the brick pool was full when the entry was claimed (the batch is refused, the host rebuilds)
struct BrickDir {
    const HashEntry* table;  // {packed brick key, brick index, unused}; nullptr: image not in brick form
    unsigned mask;           // table size - 1
    const uint2* cells;      // [n_cap][kBrickStride] {begin, count}
    unsigned n_cap;          // brick pool size (an index >= n_cap is "missing")
};
__host__ __device__ __forceinline__ unsigned brick_hash(const int bx, const int by, const int bz) {
    const unsigned h = ((unsigned)bx * 73856093u) ^ ((unsigned)by * 19349663u) ^ ((unsigned)bz * 83492791u);
    return h ^ (h >> 15);
}
// slab index of stored coordinates (0..9 per axis; interior voxel l = 0..7 sits at l + 1)
__host__ __device__ __forceinline__ unsigned brick_slab_index(const int sx, const int sy, const int sz) {
    return (unsigned)((sz * kBrickStored + sy) * kBrickStored + sx);
}
__host__ __device__ __forceinline__ bool brick_slab_interior(const unsigned l, int& sx, int& sy, int& sz) {
    sz = (int)(l / (unsigned)(kBrickStored * kBrickStored));
    const unsigned r = l - (unsigned)sz * (unsigned)(kBrickStored * kBrickStored);
    sy = (int)(r / (unsigned)kBrickStored);
    sx = (int)(r - (unsigned)sy * (unsigned)kBrickStored);
    return sz >= 1 && sz <= kBrickSide && sy >= 1 && sy <= kBrickSide && sx >= 1 && sx <= kBrickSide;  // (l >= 1000: sz >= 10 -> false)
}
// The bricks whose halo mirrors interior voxel (lx, ly, lz): f(dx, dy, dz) for every non-zero face / edge combination of the
// per-axis directions e = -1 (l == 0), +1 (l == 7), 0 (inside).  The mirrored cell sits at stored coordinate l + 1 - 8 d.
template <class F>
__host__ __device__ __forceinline__ void brick_for_each_mirror(const int lx, const int ly, const int lz, F&& f) {
    const int ex = lx == 0 ? -1 : lx == kBrickSide - 1 ? 1 : 0, ey = ly == 0 ? -1 : ly == kBrickSide - 1 ? 1 : 0, ez = lz == 0 ? -1 : lz == kBrickSide - 1 ? 1 : 0;
    if ((ex | ey | ez) == 0) return;
    for (int m = 1; m < 7; ++m) {  // (m == 7: the corner, never probed)
        if (((m & 1) && !ex) || ((m & 2) && !ey) || ((m & 4) && !ez)) continue;
        f((m & 1) ? ex : 0, (m & 2) ? ey : 0, (m & 4) ? ez : 0);
    }
}

struct GnState {
    double T[16];  // current pose, column-major 4x4 (world <- body)
    double last_rot, last_pos;
    double sum_res, sum_res2;
    double last_dx[6];
    double H[36], g[6];  // last assembled system (introspection)
    int iter;            // iterations executed so far
    int done;            // 1: stop rule hit / failed -> later launches exit at once
    int converged;       // ICP: has_converge_;  NDT: 0 if min_effective check failed
    int n_valid, n_valid2;
    int pad;
    long long dbg[16];  // cycle stamps of the solve kernel phases (diagnostics, written only with -DFLS_TIMING)
    double log_T[kMaxIter][16];
    double log_res[kMaxIter];
    int log_nv[kMaxIter];
};

// Result mailbox in host-mapped pinned memory: the Gauss-Newton tail writes the few words a Match returns
// (pose, counters) straight to the host and publishes them with one system-scope release store of `seq`
// = match_id << 9 | done << 8 | iterations.  The host spins on that word instead of paying a blocking
// hipStreamSynchronize (~25 us wake-up) plus a device-to-host copy kernel per Match.
struct Mailbox {
    double T[16];
    double sum_res, sum_res2;
    double last_dx[6];
    int iter, done, converged, n_valid, n_valid2;
    unsigned seq;
};

#ifdef __HIPCC__
// Publish an iteration result to the host-mapped mailbox without a release FENCE (a system-scope release writes
// back the whole XCD L2 first -- megabytes of freshly written Jacobian rows -- on the critical path): every
// payload word goes out as a write-through relaxed system-scope store, the wave drains its stores, then `seq`
// follows.  The host's acquire load of `seq` (matcher_base.hpp wait_mailbox) completes the hand-off.
// launch_word = max_iterations << 24 | match_id (23 bits).  The payload (pose, statistics) crosses PCIe only when the host will
// read it -- the iteration that stops the Match or the last one the host can launch; every other iteration publishes the
// sequence word alone (the host only counts iterations then) and does not wait for its stores.
__device__ __forceinline__ void mailbox_publish(Mailbox* __restrict__ mb, const double (&T)[16], const double (&dx)[6], const double sum_res,
                                                const double sum_res2, const int iter, const int done, const int converged,
                                                const int n_valid, const int n_valid2, const unsigned launch_word) {
#define FLS_MB_F64(dst, v) __hip_atomic_store((unsigned long long*)&(dst), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define FLS_MB_I32(dst, v) __hip_atomic_store(&(dst), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
    const int max_it = (int)(launch_word >> 24);
    const unsigned seq = ((launch_word & 0x7fffffu) << 9) | ((unsigned)done << 8) | (unsigned)iter;
    if (done || max_it == 0 || iter >= max_it) {
        for (int q = 0; q < 16; ++q) FLS_MB_F64(mb->T[q], T[q]);
        for (int q = 0; q < 6; ++q) FLS_MB_F64(mb->last_dx[q], dx[q]);
        FLS_MB_F64(mb->sum_res, sum_res);
        FLS_MB_F64(mb->sum_res2, sum_res2);
        FLS_MB_I32(mb->iter, iter);
        FLS_MB_I32(mb->done, done);
        FLS_MB_I32(mb->converged, converged);
        FLS_MB_I32(mb->n_valid, n_valid);
        FLS_MB_I32(mb->n_valid2, n_valid2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FLS_MB_I32(mb->seq, seq);
#undef FLS_MB_F64
#undef FLS_MB_I32
}
#endif

#ifdef __HIPCC__
// exclusive block scan of a small vector of counters (wave shuffles + one LDS hop); returns the exclusive prefix, total in `tot`
template <int NV>
__device__ __forceinline__ void block_excl_scan(unsigned (&v)[NV], unsigned (&tot)[NV], unsigned (*wsum)[NV] /* LDS [waves][NV] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned inc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        unsigned x = v[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        inc[k] = x;
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NV; ++k) wsum[w][k] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        unsigned base = 0, t = 0;
        for (int q = 0; q < nw; ++q) { const unsigned s = wsum[q][k]; if (q < w) base += s; t += s; }
        v[k] = base + inc[k] - v[k];
        tot[k] = t;
    }
    __syncthreads();
}

// sum over a small array by a whole workgroup, split at `cut`: pre = sum of v[j] for j < cut, all = sum over [0, n) (every thread gets
// both).  Used by kernels that derive their block offset from the per-block totals of the previous launch themselves instead of
// waiting for a one-workgroup scan launch in between (a dependent launch costs ~7 us on this path, the sums a fraction of one).
template <int NV, class T4>
__device__ __forceinline__ void block_prefix_total(const T4* __restrict__ v, const int n, const int cut, unsigned (&pre)[NV], unsigned (&all)[NV],
                                                   unsigned (*wsum)[2 * NV] /* LDS [waves][2 NV] */) {
    unsigned p[NV], a[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) p[k] = a[k] = 0u;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const T4 e = v[j];
        const unsigned* w = reinterpret_cast<const unsigned*>(&e);
#pragma unroll
        for (int k = 0; k < NV; ++k) { a[k] += w[k]; if (j < cut) p[k] += w[k]; }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { p[k] += __shfl_xor(p[k], o, 64); a[k] += __shfl_xor(a[k], o, 64); }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { wsum[wv][k] = p[k]; wsum[wv][NV + k] = a[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        unsigned sp = 0u, sa = 0u;
        for (int q = 0; q < nw; ++q) { sp += wsum[q][k]; sa += wsum[q][NV + k]; }
        pre[k] = sp; all[k] = sa;
    }
    __syncthreads();
}
#endif

// initial pose handed to the first iteration's kernels as a launch argument (no host-to-device copy)
struct Pose16 { double m[16]; };

struct TrafficCounters {
    unsigned long long probes, hits, cand;
};

__host__ __device__ __forceinline__ unsigned long long pack_key(int x, int y, int z) {
    return ((unsigned long long)((unsigned)x & 0x1FFFFFu) << 42) | ((unsigned long long)((unsigned)y & 0x1FFFFFu) << 21) |
           (unsigned long long)((unsigned)z & 0x1FFFFFu);
}
__host__ __device__ __forceinline__ void unpack_key(unsigned long long k, int& x, int& y, int& z) {
    x = (int)((unsigned)(k >> 42) << 11) >> 11;  // sign-extend the 21-bit fields
    y = (int)((unsigned)((k >> 21) & 0x1FFFFFu) << 11) >> 11;
    z = (int)((unsigned)(k & 0x1FFFFFu) << 11) >> 11;
}
__host__ __device__ __forceinline__ unsigned hash_key(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (unsigned)k;
}

// ---- LRU evictions inside a batch (ivox_evict_select, ndt_evict_select): creation ranks of a batch merged with the first ranks of the
// voxels it evicts and re-creates ----
// k-th (0-based) smallest of crank[0, C) U S[0, nS) (both ascending, all ranks distinct); k < C + nS
__device__ __forceinline__ unsigned evict_merged_rank(const unsigned* __restrict__ crank, const unsigned C, const unsigned* S, const unsigned nS, const unsigned k) {
    unsigned j = 0;
    while (j < nS && j <= k && S[j] < (k - j < C ? crank[k - j] : 0xFFFFFFFFu)) ++j;  // S[j] is smaller than a creation rank inside the first k + 1: it is inside too
    unsigned v = 0u;
    if (j <= k && k - j < C) v = crank[k - j];
    if (j > 0u && S[j - 1] > v) v = S[j - 1];
    return v;
}

}  // namespace fls
