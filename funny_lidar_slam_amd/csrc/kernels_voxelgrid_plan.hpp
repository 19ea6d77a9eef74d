// kernels_voxelgrid_plan.hpp -- the device VoxelGrid as ONE uninterrupted stream of launches (round 5).
//
// Rounds 2-4 read the bounds back (hipMemcpyAsync + hipStreamSynchronize) to derive the leaf grid on the host, initialised the exact
// sort's queue with three runtime fill / copy operations, and synchronised the stream again for the output size: ~90 us of a 680 us
// IncrementalNDT call were idle queue (profiles/r05_a_vg_ndt_before_sequence.txt).  Here
//   vg_minmax_plan   bounds (one write-through row per block) + ticket; the LAST block folds the rows, derives the leaf grid with the host's float arithmetic
//                    (voxel_grid.hpp:69-92: "leaf size too small" refusal, min_b / div_b / divb_mul), writes the VgPlan the following
//                    kernels read, re-arms the accumulators for the next call and initialises the exact sort's queue;
//   vg_index_plan / vg_heads_plan / vg_centroid_plan   the round-2 bodies, grid and verdict read from the plan;
//   vg_scan_publish  block offsets of the run heads + the output size, and the call's verdict {n_out, status, sort failure} published to a
//                    host-mapped mailbox: the host learns the size BEFORE the centroid kernel has run and queues the Match behind it.
// A refused call (no finite point, leaf box too large, a non-finite point in exact mode, introsort's depth limit on a range that does not
// fit into LDS) changes nothing the caller reads: every later kernel exits on plan.status, the host takes the exact host filter.
#pragma once
#include "kernels_voxelgrid.hpp"
#include "kernels_exactsort.hpp"

namespace fls {

enum : unsigned { kVgOk = 0u, kVgNoFinitePoint = 1u, kVgLeafTooSmall = 2u, kVgNonFinitePoints = 3u };

struct VgPlan {
    VgGrid g;
    unsigned status;
    unsigned n_bad;
    unsigned mn[3], mx[3];  // ordered-uint bounds (diagnostics / tests)
};
// what the blocks of vg_minmax_plan hand to their last arriver: one row of partial bounds per block (written write-through, read with sc1 loads:
// the hand-off of the fit kernels) and one ticket.  (Until late in round 5 every block folded its bounds into six shared words with device-scope
// atomics: 160 blocks x 6 atomics on six addresses serialise at the memory side -- 11 us for a kernel that reads 1.4 MB.)
constexpr int kVgMinmaxMaxBlocks = 512;
struct VgAccum { unsigned ticket, pad[7]; unsigned part[kVgMinmaxMaxBlocks][8]; };  // part[b] = {mn x y z, mx x y z, n_bad, -}
// host-mapped: the verdict of one call
struct VgMailbox { unsigned seq, n_out, status, sort_fail; };
// what the last block of vg_minmax_plan initialises for es_task_kernel (st == nullptr: no exact sort follows)
struct EsInitArgs { EsState* st; EsQueue* q; unsigned* ready; unsigned work_cap; };

__device__ __forceinline__ float vg_unord_dev(const unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void __launch_bounds__(kVgBlock)
vg_minmax_plan(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const float inv, const int refuse_bad,
               VgAccum* __restrict__ acc, VgPlan* __restrict__ plan, const EsInitArgs es) {
    __shared__ unsigned red[kVgBlock / 64][6], s_bad[kVgBlock / 64];
    __shared__ unsigned s_last;
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
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        unsigned b2 = bad;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b2 += __shfl_xor(b2, o, 64);
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { red[w][a] = lo[a]; red[w][3 + a] = hi[a]; }
            s_bad[w] = b2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        unsigned v;
        if (a < 6) {
            v = red[0][a];
            for (int q = 1; q < kVgBlock / 64; ++q) v = a < 3 ? min(v, red[q][a]) : max(v, red[q][a]);
        } else {
            v = 0u;
            for (int q = 0; q < kVgBlock / 64; ++q) v += s_bad[q];
        }
        __hip_atomic_store(&acc->part[blockIdx.x][a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
    }
    // the block's row has left the chip's caches (drained) before its ticket is taken
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&acc->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // last block: fold the rows (one per thread, sc1 loads), then one thread derives the plan
    {
        unsigned r[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
        for (unsigned bq = threadIdx.x; bq < gridDim.x; bq += kVgBlock) {
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                const unsigned v = __hip_atomic_load(&acc->part[bq][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r[a] = a < 3 ? min(r[a], v) : a < 6 ? max(r[a], v) : r[a] + v;
            }
        }
#pragma unroll
        for (int a = 0; a < 7; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned t2 = (unsigned)__shfl_xor((int)r[a], o, 64);
                r[a] = a < 3 ? min(r[a], t2) : a < 6 ? max(r[a], t2) : r[a] + t2;
            }
        }
        __syncthreads();  // (red / s_bad are reused)
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) red[w][a] = r[a];
            s_bad[w] = r[6];
        }
        __syncthreads();
    }
    if (es.st != nullptr) {  // the exact sort's queue: the whole array is workgroup 0's first task (open = 1 stands for it)
        for (unsigned i = threadIdx.x; i < es.work_cap; i += kVgBlock) es.ready[i] = 0u;
        if (threadIdx.x < sizeof(EsState) / 4u) reinterpret_cast<unsigned*>(es.st)[threadIdx.x] = 0u;
        if (threadIdx.x == 0) *es.q = EsQueue{0u, 0u, 1u, 0u};
    }
    if (threadIdx.x != 0) return;
    unsigned umn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, umx[3] = {0u, 0u, 0u}, nbad = 0u;
    for (int q = 0; q < kVgBlock / 64; ++q) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { umn[a] = min(umn[a], red[q][a]); umx[a] = max(umx[a], red[q][3 + a]); }
        nbad += s_bad[q];
    }
    VgPlan p;
    p.n_bad = nbad;
#pragma unroll
    for (int a = 0; a < 3; ++a) { p.mn[a] = umn[a]; p.mx[a] = umx[a]; }
    p.g = VgGrid{inv, {0, 0, 0}, 0, 0, 0u};
    p.status = kVgOk;
    if (umn[0] == 0xffffffffu) p.status = kVgNoFinitePoint;
    else if (refuse_bad && nbad != 0u) p.status = kVgNonFinitePoints;
    else {
        float mn[3], mx[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = vg_unord_dev(umn[a]); mx[a] = vg_unord_dev(umx[a]); }
        // voxel_grid.hpp:69-74: dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inverse_leaf_size_[0]) + 1, refuse if dx dy dz > INT_MAX
        const float ex = (mx[0] - mn[0]) * inv, ey = (mx[1] - mn[1]) * inv, ez = (mx[2] - mn[2]) * inv;
        // one extent of 2^31 leaves or more already makes dx dy dz > INT_MAX: refuse there, and every product below stays inside int64
        // (the former bound 4e9 let dx dy wrap for two extents of ~3.1e9 leaves)
        const float lim = 2147483648.0f;
        bool too_small = !(ex < lim && ey < lim && ez < lim);
        if (!too_small) {
            const long long dx = (long long)ex + 1, dy = (long long)ey + 1, dz = (long long)ez + 1;
            const long long dxy = dx * dy;  // <= (2^31 + 1)^2 < 2^63
            too_small = dxy > 0x7fffffffLL || dxy * dz > 0x7fffffffLL;
        }
        if (too_small) p.status = kVgLeafTooSmall;
        else {
            long long div_b[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                p.g.min_b[a] = (int)floorf(mn[a] * inv);
                div_b[a] = (long long)(int)floorf(mx[a] * inv) - p.g.min_b[a] + 1;
            }
            const long long total = div_b[0] * div_b[1] * div_b[2];
            if (total <= 0 || total >= 0x7fffffffLL) p.status = kVgLeafTooSmall;
            else {
                p.g.m1 = (int)div_b[0];
                p.g.m2 = (int)(div_b[0] * div_b[1]);
                p.g.total = (unsigned)total;
            }
        }
    }
    *plan = p;
    __hip_atomic_store(&acc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next call (this block is the only one left; the rows are simply overwritten)
}

__global__ void __launch_bounds__(kVgBlock)
vg_index_plan(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const VgPlan* __restrict__ plan,
              unsigned* __restrict__ key, unsigned* __restrict__ val) {
    if (plan->status != kVgOk) return;
    const VgGrid g = plan->g;
    vg_index_body(x, y, z, n, g, key, val);
}
__global__ void __launch_bounds__(kVgScanBlock)
vg_heads_plan(const unsigned* __restrict__ key, const unsigned* __restrict__ val, const int n, const VgPlan* __restrict__ plan, const EsState* __restrict__ st,
              const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ in, float4* __restrict__ sorted,
              unsigned* __restrict__ lx, unsigned* __restrict__ bt) {
    if (plan->status != kVgOk || (st != nullptr && st->fail != 0u)) {  // (a failed sort leaves a permutation of the input, but nobody reads it)
        if (threadIdx.x == 0) bt[blockIdx.x] = 0u;
        return;
    }
    vg_heads_body(key, val, n, plan->g.total, x, y, z, in, sorted, lx, bt);
}
__global__ void __launch_bounds__(kVgScanBlock)
vg_scan_publish(const unsigned* __restrict__ in, unsigned* __restrict__ out, const int m, const VgPlan* __restrict__ plan, const EsState* __restrict__ st,
                VgMailbox* __restrict__ mb, const unsigned seq) {
    const unsigned carry = vg_scan_body(in, out, m);
    if (threadIdx.x != 0) return;
    const unsigned fail = st != nullptr ? st->fail : 0u;
    __hip_atomic_store(&mb->n_out, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->status, plan->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->sort_fail, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(kVgBlock)
vg_centroid_plan(const unsigned* __restrict__ key, const float4* __restrict__ sorted, const int n, const VgPlan* __restrict__ plan, const EsState* __restrict__ st,
                 const unsigned* __restrict__ lx, const unsigned* __restrict__ bt, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                 float* __restrict__ oi) {
    if (plan->status != kVgOk || (st != nullptr && st->fail != 0u)) return;
    vg_centroid_body(key, sorted, n, lx, bt, ox, oy, oz, oi);
}

}  // namespace fls
