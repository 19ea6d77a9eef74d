// kernels_voxelgrid.hpp -- pcl::VoxelGrid<PointXYZI>::filter on the device.
//
// Reference: VoxelGridCloud (include/common/pointcloud_utility.h:216-271) -> pcl::VoxelGrid (PCL 1.10 voxel_grid.hpp):
// finite bounds, leaf index i + j*dx + k*dx*dy per finite point, std::sort of (index, point) records by index, one
// centroid per run of equal indices, runs in ascending index order.
//
// CONTRACT (why this path is opt-in, FLS_DEVICE_VOXELGRID=1):
//   * the SET of leaves, their ORDER and every integer in the pipeline are the reference's (bit-exact);
//   * a centroid is the float sum of the leaf's points divided by their count.  The reference sums in the order
//     libstdc++'s std::sort (introsort, unstable) leaves equal keys in; that order is not a function of the leaf alone, so
//     no parallel algorithm reproduces it.  Here a leaf is summed in ASCENDING POINT INDEX (a stable radix sort): leaves
//     holding one or two points are bit-identical (float addition commutes), larger leaves differ in the last bits of the
//     float sum (|delta| <= (count - 2) ulp of the running sum per component; tests/test_gpu_voxelgrid.py measures it).
//   The host path (host_maps.hpp voxel_grid, the exact std::sort order) stays the default.
//
// Kernels (n points, 256-thread blocks, tiles of 1024 keys):
//   vg_minmax      finite bounds (ordered-uint atomics, one set per block)
//   vg_index       leaf index per point (non-finite points get the sentinel `total`, sorted last and dropped)
//   vg_hist / vg_scan_rows / vg_scatter   one stable LSD radix pass over 8 bits (ceil(bits(total) / 8) passes)
//   vg_heads / vg_scan / vg_centroid      run heads (+ points gathered into sorted order) -> output slot -> sequential float sums
#pragma once
#include "device_common.hpp"

namespace fls {

constexpr int kVgBlock = 256;
constexpr int kVgItems = 4;
constexpr int kVgTile = kVgBlock * kVgItems;  // keys per block of a radix pass
constexpr int kVgMaxBlocks = 4096;            // n <= 4,194,304 (vg_scan_rows walks its row in chunks of 1024 tiles)
constexpr int kVgScanBlock = 1024;

struct VgHeader {
    unsigned mn[3], mx[3];  // ordered-uint encoded float bounds of the finite points
    unsigned n_out;
    unsigned n_bad;  // points with a non-finite coordinate (vg_minmax)
};

__device__ __forceinline__ unsigned vg_ord(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float vg_unord(unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
__device__ __forceinline__ bool vg_finite3(float x, float y, float z) {
    return fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY;  // false for NaN
}

__global__ void __launch_bounds__(kVgBlock)
vg_minmax(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, VgHeader* __restrict__ h) {
    __shared__ unsigned red[kVgBlock / 64][6];
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, bad = 0u;
    for (int i = blockIdx.x * kVgBlock + threadIdx.x; i < n; i += gridDim.x * kVgBlock) {
        const float px = x[i], py = y[i], pz = z[i];
        if (!vg_finite3(px, py, pz)) { ++bad; continue; }
        const unsigned ox = vg_ord(px), oy = vg_ord(py), oz = vg_ord(pz);
        lo[0] = min(lo[0], ox); hi[0] = max(hi[0], ox);
        lo[1] = min(lo[1], oy); hi[1] = max(hi[1], oy);
        lo[2] = min(lo[2], oz); hi[2] = max(hi[2], oz);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], o, 64));
            hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], o, 64));
        }
    }
    if (bad) atomicAdd(&h->n_bad, bad);  // non-finite points (rare): the cell-grid build refuses such a cloud like the host build does
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[w][a] = lo[a]; red[w][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        unsigned v = red[0][a];
        for (int q = 1; q < kVgBlock / 64; ++q) v = a < 3 ? min(v, red[q][a]) : max(v, red[q][a]);
        if (a < 3) atomicMin(&h->mn[a], v);
        else atomicMax(&h->mx[a - 3], v);
    }
}

struct VgGrid {
    float inv;
    int min_b[3];
    int m1, m2;
    unsigned total;  // number of leaves of the box = the sentinel key of a non-finite point
};

__device__ __forceinline__ void vg_index_body(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const VgGrid& g,
                                              unsigned* __restrict__ key, unsigned* __restrict__ val) {
    const int i = blockIdx.x * kVgBlock + threadIdx.x;
    if (i >= n) return;
    const float px = x[i], py = y[i], pz = z[i];
    unsigned k = g.total;
    if (vg_finite3(px, py, pz)) {
        // voxel_grid.hpp:  static_cast<int>(std::floor(p.x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0]))
        const int i0 = (int)(floorf(px * g.inv) - (float)g.min_b[0]);
        const int i1 = (int)(floorf(py * g.inv) - (float)g.min_b[1]);
        const int i2 = (int)(floorf(pz * g.inv) - (float)g.min_b[2]);
        k = (unsigned)(i0 + i1 * g.m1 + i2 * g.m2);
    }
    key[i] = k;
    val[i] = (unsigned)i;
}
__global__ void __launch_bounds__(kVgBlock)
vg_index(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const VgGrid g,
         unsigned* __restrict__ key, unsigned* __restrict__ val) {
    vg_index_body(x, y, z, n, g, key, val);
}

// per-tile digit counts, digit-major: hist[d * nb + b]
__global__ void __launch_bounds__(kVgBlock)
vg_hist(const unsigned* __restrict__ key, const int n, const int shift, unsigned* __restrict__ hist, const int nb, unsigned* __restrict__ dig_tot) {
    __shared__ unsigned c[256];
    c[threadIdx.x] = 0u;
    __syncthreads();
    const int base = blockIdx.x * kVgTile;
#pragma unroll
    for (int r = 0; r < kVgItems; ++r) {
        const int e = base + r * kVgBlock + threadIdx.x;
        if (e < n) atomicAdd(&c[(key[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nb + blockIdx.x] = c[threadIdx.x];
    if (c[threadIdx.x]) atomicAdd(&dig_tot[threadIdx.x], c[threadIdx.x]);  // integer sums: order-free
}

// exclusive scan of the digit-major histogram, one workgroup per digit: base = keys of all smaller digits (from the digit
// totals vg_hist accumulated), then the scan of this digit's row of nb tile counts.  (One workgroup over all 256 x nb
// counters was a 17-35 us serial chain; this is 256-way parallel.)
__global__ void __launch_bounds__(kVgScanBlock)
vg_scan_rows(const unsigned* __restrict__ hist, unsigned* __restrict__ out, const int nb, const unsigned* __restrict__ dig_tot) {
    __shared__ unsigned wsum[kVgScanBlock / 64], bsum[kVgScanBlock / 64];
    const int d = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned lower = ((int)threadIdx.x < d && threadIdx.x < 256u) ? dig_tot[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lower += __shfl_xor(lower, o, 64);
    if (lane == 0) bsum[w] = lower;
    __syncthreads();
    unsigned base = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) base += bsum[q];
    // rows longer than one workgroup (more than 1024 tiles = 1,048,576 keys: the map-side filters of the LOAM deques) go in
    // chunks of 1024 tile counters with a carried running total
    unsigned carry = 0u;
    for (int b0 = 0; b0 < nb; b0 += kVgScanBlock) {
        const int idx = b0 + (int)threadIdx.x;
        const unsigned v = idx < nb ? hist[d * nb + idx] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        unsigned wbase = 0u, tot = 0u;
#pragma unroll
        for (int q = 0; q < kVgScanBlock / 64; ++q) { const unsigned t = wsum[q]; if (q < w) wbase += t; tot += t; }
        if (idx < nb) out[d * nb + idx] = base + carry + wbase + inc - v;
        carry += tot;
        __syncthreads();  // wsum is rewritten by the next chunk
    }
}

// one workgroup: exclusive scan of `m` counters, in -> out, in coalesced tiles of 1024 with a carried running total.  The
// loads of 32 tiles are issued together (one ~1 us memory round trip per 32 tiles instead of one per tile); a contiguous
// chunk per thread made every wave load touch 64 cache lines: 27-35 us on the one CU that runs this.
__device__ __forceinline__ unsigned vg_scan_body(const unsigned* __restrict__ in, unsigned* __restrict__ out, const int m) {
    __shared__ unsigned wsum[2][kVgScanBlock / 64];
    constexpr int B = 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned carry = 0u;
    for (int sb = 0; sb < m; sb += B * kVgScanBlock) {
        unsigned v[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int i = sb + k * kVgScanBlock + (int)threadIdx.x;
            v[k] = i < m ? in[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int base = sb + k * kVgScanBlock;
            if (base < m) {  // uniform
                const int i = base + (int)threadIdx.x;
                unsigned inc = v[k];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
                if (lane == 63) wsum[k & 1][w] = inc;
                __syncthreads();  // one barrier per tile: the LDS row alternates, a row is rewritten two tiles later
                unsigned wbase = 0u, tot = 0u;
#pragma unroll
                for (int q = 0; q < kVgScanBlock / 64; ++q) { const unsigned t = wsum[k & 1][q]; if (q < w) wbase += t; tot += t; }
                if (i < m) out[i] = carry + wbase + inc - v[k];
                carry += tot;
            }
        }
    }
    return carry;
}
__global__ void __launch_bounds__(kVgScanBlock)
vg_scan(const unsigned* __restrict__ in, unsigned* __restrict__ out, const int m, unsigned* __restrict__ total_out) {
    const unsigned carry = vg_scan_body(in, out, m);
    if (total_out != nullptr && threadIdx.x == 0) *total_out = carry;
}

// stable scatter of one tile: destination = scanned histogram entry + keys of the same digit earlier in the tile.
// Tile order is (round r, thread t) = key index; the in-tile rank is wave-match (8 ballots) + an LDS prefix over (r, wave).
__global__ void __launch_bounds__(kVgBlock)
vg_scatter(const unsigned* __restrict__ kin, const unsigned* __restrict__ vin, unsigned* __restrict__ kout, unsigned* __restrict__ vout,
           const int n, const int shift, const unsigned* __restrict__ hist, const int nb) {
    constexpr int kW = kVgBlock / 64;
    __shared__ unsigned cnt[kVgItems][kW][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int q = t; q < kVgItems * kW * 256; q += kVgBlock) (&cnt[0][0][0])[q] = 0u;
    __syncthreads();
    unsigned key[kVgItems], val[kVgItems], lrank[kVgItems];
    const int base = blockIdx.x * kVgTile;
#pragma unroll
    for (int r = 0; r < kVgItems; ++r) {
        const int e = base + r * kVgBlock + t;
        const bool valid = e < n;
        key[r] = valid ? kin[e] : 0u;
        val[r] = valid ? vin[e] : 0u;
        const unsigned d = (key[r] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool on = (d >> bit) & 1u;
            const unsigned long long bm = __ballot(on);
            m &= on ? bm : ~bm;
        }
        lrank[r] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (valid && lrank[r] == 0u) cnt[r][w][d] = (unsigned)__popcll(m);
    }
    __syncthreads();
    {
        unsigned run = 0u;
#pragma unroll
        for (int r = 0; r < kVgItems; ++r)
#pragma unroll
            for (int q = 0; q < kW; ++q) { const unsigned c = cnt[r][q][t]; cnt[r][q][t] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kVgItems; ++r) {
        const int e = base + r * kVgBlock + t;
        if (e >= n) continue;
        const unsigned d = (key[r] >> shift) & 255u;
        const unsigned dst = hist[d * nb + blockIdx.x] + cnt[r][w][d] + lrank[r];
        kout[dst] = key[r];
        vout[dst] = val[r];
    }
}

// run heads of the sorted keys: block-local exclusive rank + block totals
__device__ __forceinline__ void vg_heads_body(const unsigned* __restrict__ key, const unsigned* __restrict__ val, const int n, const unsigned total, const float* __restrict__ x,
                                              const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ in, float4* __restrict__ sorted,
                                              unsigned* __restrict__ lx, unsigned* __restrict__ bt) {
    __shared__ unsigned wsum[kVgScanBlock / 64];
    const int e = blockIdx.x * kVgScanBlock + threadIdx.x;
    unsigned head = 0u;
    if (e < n) {
        const unsigned k = key[e];
        head = (k < total && (e == 0 || key[e - 1] != k)) ? 1u : 0u;
        const unsigned p = val[e];
        sorted[e] = make_float4(x[p], y[p], z[p], in[p]);  // the points in sorted order: the centroid walk reads contiguous memory
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = head;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0u, tot = 0u;
    for (int q = 0; q < kVgScanBlock / 64; ++q) { const unsigned t = wsum[q]; if (q < w) base += t; tot += t; }
    if (e < n) lx[e] = head ? (base + inc - 1u) : 0xffffffffu;
    if (threadIdx.x == 0) bt[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kVgScanBlock)
vg_heads(const unsigned* __restrict__ key, const unsigned* __restrict__ val, const int n, const unsigned total, const float* __restrict__ x,
         const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ in, float4* __restrict__ sorted,
         unsigned* __restrict__ lx, unsigned* __restrict__ bt) {
    vg_heads_body(key, val, n, total, x, y, z, in, sorted, lx, bt);
}

// one thread per run head: float sums in sorted (= ascending point index) order, centroid = sum / count.
// Round 6: a run of kVgLongLeaf points or more is summed by the head's WAVE instead of its one thread -- sixty-four points per round arrive as one
// coalesced load (the next round's already in flight), then the sums are folded in index order through v_readlane: the same chain of IEEE additions
// ((s + p0) + p1) + ..., ~19 ns per point instead of ~125 (one thread with eight 16-byte loads in flight).  The map-side filters of the kd-tree kinds
// see leaves of a thousand points (the keyframe deque near the sensor): vg_centroid 137 us on the 1.55 M-point planar deque of LoamFull before.
constexpr int kVgLongLeaf = 128;
__device__ __forceinline__ void vg_centroid_body(const unsigned* __restrict__ key, const float4* __restrict__ sorted, const int n, const unsigned* __restrict__ lx,
                                                 const unsigned* __restrict__ bt, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, float* __restrict__ oi) {
    const int e = blockIdx.x * kVgBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned l = e < n ? lx[e] : 0xffffffffu;
    const bool head = l != 0xffffffffu;
    unsigned o = 0u;
    int hi = e, c = 0;
    if (head) {
        o = bt[e / kVgScanBlock] + l;
        const unsigned k = key[e];
        // end of the run: gallop, then bisect (the keys are sorted) -- ~2 log2(length) dependent loads instead of one per point
        // (round 3: the walk that tested key[q] before every add was latency-bound, 130-520 us on leaves of hundreds of points)
        int lo = e, step = 1;  // key[lo] == k
        while (lo + step < n && key[lo + step] == k) { lo += step; step <<= 1; }
        hi = min(lo + step, n);  // key[hi] != k or hi == n
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (key[mid] == k) lo = mid; else hi = mid; }
        c = hi - e;
    }
    const bool long_run = head && c >= kVgLongLeaf;
    if (head && !long_run) {
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int q = e;
        for (; q + 8 <= hi; q += 8) {  // eight loads in flight, adds in index order
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = sorted[q + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { sx += p[u].x; sy += p[u].y; sz += p[u].z; si += p[u].w; }
        }
        for (; q < hi; ++q) {
            const float4 p = sorted[q];
            sx += p.x; sy += p.y; sz += p.z; si += p.w;
        }
        const float cf = (float)c;
        ox[o] = __fdiv_rn(sx, cf);
        oy[o] = __fdiv_rn(sy, cf);
        oz[o] = __fdiv_rn(sz, cf);
        oi[o] = __fdiv_rn(si, cf);
    }
    // the long runs of this wave, one after the other, by all of its lanes (wave-uniform control flow)
    unsigned long long pending = __ballot(long_run);
    while (pending != 0ull) {
        const int src = __ffsll((long long)pending) - 1;
        pending &= pending - 1ull;
        const int b = __shfl(e, src, 64), end = __shfl(hi, src, 64);
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        float4 nxt = (b + lane < end) ? sorted[b + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = b; q < end; q += 64) {
            const float4 p = nxt;
            if (q + 64 < end) nxt = (q + 64 + lane < end) ? sorted[q + 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int cnt = min(64, end - q);
            if (cnt == 64) {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    sx += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(p.x), j));
                    sy += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(p.y), j));
                    sz += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(p.z), j));
                    si += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(p.w), j));
                }
            } else {
                for (int j = 0; j < cnt; ++j) {
                    sx += __shfl(p.x, j, 64); sy += __shfl(p.y, j, 64); sz += __shfl(p.z, j, 64); si += __shfl(p.w, j, 64);
                }
            }
        }
        if (lane == src) {
            const float cf = (float)c;
            ox[o] = __fdiv_rn(sx, cf);
            oy[o] = __fdiv_rn(sy, cf);
            oz[o] = __fdiv_rn(sz, cf);
            oi[o] = __fdiv_rn(si, cf);
        }
    }
}
__global__ void __launch_bounds__(kVgBlock)
vg_centroid(const unsigned* __restrict__ key, const float4* __restrict__ sorted, const int n, const unsigned* __restrict__ lx,
            const unsigned* __restrict__ bt, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, float* __restrict__ oi) {
    vg_centroid_body(key, sorted, n, lx, bt, ox, oy, oz, oi);
}

}  // namespace fls
