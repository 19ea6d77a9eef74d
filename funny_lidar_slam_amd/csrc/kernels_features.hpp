// kernels_features.hpp -- LOAM feature front-end on the device: the step right before Match in
// LoamFull_KdTree mode.
//   loam::PointcloudProjector::Project       src/loam/pointcloud_projector.cpp:32-133   feat_project / feat_count / feat_compact
//   loam::FeatureExtractor::SelectValidPoints src/loam/feature_extractor.cpp:65-117     feat_valid_rough_kernel
//   loam::FeatureExtractor::ComputeRoughness  :48-62                                     feat_valid_rough_kernel
//   loam::FeatureExtractor::SelectFeatures    :119-222                                   feat_select_kernel
// All float arithmetic is the reference's sequence (no FMA contraction; correctly rounded sqrt and division), integer
// results (cell owners, ordered indices, flags, selections) are bit-exact against the oracle.
//
// Parallel structure: the range image is filled by "first point in stream order wins" = atomicMin of the raw index per
// cell; rows compact independently (ballot prefix sums); the per-point marks are idempotent stores of 0, so they
// commute; the greedy selection is sequential inside one ring (the six sectors of a ring share the valid flags and
// one boundary element) but independent across rings: one wave per ring, bitonic sector sort in LDS (key =
// {roughness bits : position} = the stable order the oracle fixes for std::sort's unspecified ties), lane 0 walks.
#pragma once
#include "device_common.hpp"

namespace fls {

struct FeatParamsDev {
    int rows, cols;
    float h_res, min_dist, max_dist, corner_thr, planar_thr;
};
struct RawLayoutDev { unsigned stride, off_xyz, off_i, off_ring; };

constexpr unsigned kFeatNone = 0xFFFFFFFFu;
constexpr int kFeatMaxSector = 1024;  // elements per sector the LDS sort holds: (cols - 11) / 6 <= 1024
constexpr int kFeatMaxCols = 4096;    // columns per ring the LDS row image holds

// correctly rounded float sqrt / division whatever the compiler's fp32 accuracy options: evaluate in double and round once
// (53 >= 2 * 24 + 2 bits, so the double rounding is innocuous for sqrt and for the quotient of two floats)
__device__ __forceinline__ float sqrt_rn(const float s) { return (float)sqrt((double)s); }
__device__ __forceinline__ float div_rn(const float a, const float b) { return (float)((double)a / (double)b); }

// include/common/math_function.h:159-186, Type = float
__device__ __forceinline__ float fast_atan2f_dev(const float y, const float x) {
    const float p1 = (float)0.9997878412794807, p3 = (float)-0.3258083974640975, p5 = (float)0.1555786518463281,
                p7 = (float)-0.04432655554792128;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = 1.1920928955078125e-07f;  // numeric_limits<float>::epsilon()
    float a;
    if (ax >= ay) {
        const float c = div_rn(ay, ax + eps), c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        const float c = div_rn(ax, ay + eps), c2 = c * c;
        a = (float)1.57079632679489661923 - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = (float)3.14159265358979323846 - a;
    if (y < 0) a = (float)(2 * 3.14159265358979323846) - a;
    if (a > (float)3.14159265358979323846) a -= (float)(2 * 3.14159265358979323846);
    return a;
}

__device__ __forceinline__ void raw_point(const unsigned char* __restrict__ raw, const RawLayoutDev L, const unsigned k, float& x, float& y,
                                          float& z, int& ring) {
    const unsigned char* q = raw + (size_t)k * L.stride;
    x = *(const float*)(q + L.off_xyz);
    y = *(const float*)(q + L.off_xyz + 4);
    z = *(const float*)(q + L.off_xyz + 8);
    ring = (int)*(const unsigned short*)(q + L.off_ring);
}
__device__ __forceinline__ float depth_of(const float x, const float y, const float z) { return sqrt_rn(x * x + y * y + z * z); }  // :64

// Project, first half (:58-112): range-image cell of every raw point, first in stream order wins
__global__ void __launch_bounds__(256)
feat_project_kernel(const unsigned char* __restrict__ raw, const unsigned n, const RawLayoutDev L, const FeatParamsDev p, unsigned* __restrict__ owner) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    float x, y, z;
    int row;
    raw_point(raw, L, k, x, y, z, row);
    const float d = depth_of(x, y, z);
    if (d < p.min_dist || d > p.max_dist) return;  // :66-68
    int col = (int)roundf(div_rn(fast_atan2f_dev(y, x), p.h_res)) + p.cols / 2;  // :69-70
    if (col >= p.cols) col -= p.cols;
    if (row >= p.rows || row < 0 || col < 0 || col >= p.cols) return;  // :87-88
    atomicMin(&owner[(size_t)row * p.cols + col], k);
}

// occupied cells per ring
__global__ void __launch_bounds__(256)
feat_count_kernel(const unsigned* __restrict__ owner, const FeatParamsDev p, int* __restrict__ row_count) {
    __shared__ int wsum[4];
    const int row = blockIdx.x;
    int c = 0;
    for (int col = threadIdx.x; col < p.cols; col += 256) c += owner[(size_t)row * p.cols + col] != kFeatNone ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) row_count[row] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Project, second half (:114-132): ordered cloud in (ring, column) order + depth / column / ring bounds; valid flags start true
__global__ void __launch_bounds__(256)
feat_compact_kernel(const unsigned char* __restrict__ raw, const RawLayoutDev L, const unsigned* __restrict__ owner, const FeatParamsDev p,
                    const int* __restrict__ row_count, float4* __restrict__ ordered, float* __restrict__ depth, int* __restrict__ colv,
                    int* __restrict__ raw_index, unsigned char* __restrict__ valid, int* __restrict__ row_start, int* __restrict__ row_end,
                    int* __restrict__ n_ordered) {
    __shared__ int wsum[4];
    __shared__ int s_base;
    const int row = blockIdx.x;
    if (threadIdx.x == 0) {
        int b = 0;
        for (int r = 0; r < row; ++r) b += row_count[r];
        s_base = b;
        row_start[row] = b + 5;                 // :116
        row_end[row] = b + row_count[row] - 6;  // :131
        if (row == p.rows - 1) *n_ordered = b + row_count[row];
    }
    __syncthreads();
    int run = s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < p.cols; c0 += 256) {
        const int col = c0 + (int)threadIdx.x;
        const unsigned own = col < p.cols ? owner[(size_t)row * p.cols + col] : kFeatNone;
        const bool occ = own != kFeatNone;
        const unsigned long long m = __ballot(occ);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int chunk = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (occ) {
            const int i = run + wbase + before;
            float x, y, z;
            int ring;
            raw_point(raw, L, own, x, y, z, ring);
            const float it = *(const float*)(raw + (size_t)own * L.stride + L.off_i);
            ordered[i] = make_float4(x, y, z, it);  // identity de-skew (:101-111)
            depth[i] = depth_of(x, y, z);
            colv[i] = col;
            raw_index[i] = (int)own;
            valid[i] = 1;
        }
        run += chunk;
        __syncthreads();
    }
}

// SelectValidPoints + ComputeRoughness: one lane per ordered index
__global__ void __launch_bounds__(256)
feat_valid_rough_kernel(const int* __restrict__ n_ordered, const float* __restrict__ d, const int* __restrict__ colv, float* __restrict__ rough,
                        unsigned char* __restrict__ valid) {
    const int N = *n_ordered;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (N < 12 || i >= N) return;
    if (i < 5 || i >= N - 6) valid[i] = 0;  // :68-79
    float r2 = 0.f;
    if (i >= 5 && i < N - 5) {  // :54-61, left-to-right float sum
        const float r = d[i - 5] + d[i - 4] + d[i - 3] + d[i - 2] + d[i - 1] + d[i + 1] + d[i + 2] + d[i + 3] + d[i + 4] + d[i + 5] - 10.0f * d[i];
        r2 = r * r;
    }
    rough[i] = r2;
    if (i >= 5 && i < N - 6) {  // :82-116; the marks are stores of 0, any order gives the same flags
        const float d1 = d[i], d2 = d[i + 1];
        const int cd = abs(colv[i + 1] - colv[i]);
        if (cd < 10) {
            if ((double)(d1 - d2) > 0.3) {
#pragma unroll
                for (int k = 0; k <= 5; ++k) valid[i - k] = 0;
            } else if ((double)(d2 - d1) > 0.3) {
#pragma unroll
                for (int k = 1; k <= 6; ++k) valid[i + k] = 0;
            }
        }
        const float f1 = fabsf(d[i - 1] - d1), f2 = fabsf(d2 - d1);
        if ((double)f1 > 0.02 * (double)d1 && (double)f2 > 0.02 * (double)d1) valid[i] = 0;
    }
}

// SelectFeatures (:119-222): one wave per ring
struct FeatSelectSmem {
    unsigned long long key[kFeatMaxSector];
    float rough[kFeatMaxCols];
    unsigned short col[kFeatMaxCols];
    unsigned char valid[kFeatMaxCols];
    unsigned char corner[kFeatMaxCols];
};

constexpr int kFeatSelectThreads = 256;
__global__ void __launch_bounds__(kFeatSelectThreads)
feat_select_kernel(const int* __restrict__ n_ordered, const FeatParamsDev p, const int* __restrict__ row_start, const int* __restrict__ row_end,
                   const float* __restrict__ rough, const int* __restrict__ colv, unsigned char* __restrict__ valid /* in: pre, out: post */,
                   unsigned char* __restrict__ is_corner, int* __restrict__ corner_idx /* [rows][120] */, int* __restrict__ corner_cnt,
                   int* __restrict__ planar_idx /* [rows][cols + 6] */, int* __restrict__ planar_cnt) {
    __shared__ FeatSelectSmem sm;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;
    const int N = *n_ordered;
    const int base = row_start[row] - 5, count = row_end[row] + 6 - base;  // the ring's slice of the ordered cloud
    if (N < 12) {
        if (tid == 0) { corner_cnt[row] = 0; planar_cnt[row] = 0; }
        for (int k = tid; k < count; k += kFeatSelectThreads) is_corner[base + k] = 0;
        return;
    }
    for (int k = tid; k < count; k += kFeatSelectThreads) {
        sm.rough[k] = rough[base + k];
        sm.col[k] = (unsigned short)colv[base + k];
        sm.valid[k] = valid[base + k];
        sm.corner[k] = 0;
    }
    __syncthreads();
    int nc = 0, np = 0;  // wave-0 uniform
    int* const crow = corner_idx + (size_t)row * 120;
    int* const prow = planar_idx + (size_t)row * (p.cols + 6);
    const int t = (row_end[row] - row_start[row]) / 6;  // :124-125, truncating division
    // +-5 suppression around a local position (the column-gap rule, :167-186 / :194-212).  Executed by the whole of
    // wave 0 with uniform arguments: every lane stores the same zeros to the same LDS bytes.
    // Lane-parallel: lanes 0..10 hold the columns of q-5..q+5 (one LDS read), the ten column gaps become a ballot mask,
    // the two break positions are bit scans of it, and the zeros go out as one predicated store (the scalar form of
    // the same rule cost about 1,500 cycles per call in dependent LDS round trips and branches).
    // Returns the cleared range [lo, hi] so that a caller holding prefetched flags can patch them.
    auto suppress = [&](const int q, int& lo, int& hi) {
        const int l = lane < 11 ? lane : 10;                                         // q - 5 >= 0 and q + 5 < count for every walked position
        const int c = (int)sm.col[q - 5 + l];
        const int cprev = __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, false);  // row_shr:1: the column of position q - 6 + l
        const bool gap = lane >= 1 && lane < 11 && abs(c - cprev) > 10;              // bit l: gap between positions q-6+l and q-5+l
        const unsigned gm = (unsigned)__ballot(gap);
        const unsigned fwd = gm >> 6;      // bit k-1: gap between q+k-1 and q+k, k = 1..5
        const unsigned bwd = gm & 0x3Eu;   // bit 6-k: gap between q-k and q-k+1, k = 1..5
        int nf = fwd ? __builtin_ctz(fwd) : 5;
        nf = nf < 5 ? nf : 5;
        const int nb = bwd ? 5 - (31 - __builtin_clz(bwd)) : 5;
        const int k = lane - 5;
        if (lane < 11 && k >= -nb && k <= nf) sm.valid[q + k] = 0;
        lo = q - nb;
        hi = q + nf;
    };
    for (int s = 0; s < 6; ++s) {
        const int b0 = 5 + s * t, b1 = 5 + (s + 1) * t;  // local positions of block_start_index / block_end_index
        if (b0 >= b1) continue;                          // block-uniform
        const int len = b1 - b0;
        int M = 1;
        while (M < len) M <<= 1;
        // sort the sector by {roughness, position}: bitonic network in LDS, the whole workgroup
        for (int k = tid; k < M; k += kFeatSelectThreads)
            sm.key[k] = k < len ? (((unsigned long long)__float_as_uint(sm.rough[b0 + k]) << 32) | (unsigned)(b0 + k)) : ~0ull;
        __syncthreads();
        for (int kk = 2; kk <= M; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int e = tid; e < (M >> 1); e += kFeatSelectThreads) {
                    const int lo = ((e / j) * 2 * j) + (e % j), hi = lo + j;
                    const bool up = ((lo & kk) == 0);
                    const unsigned long long a = sm.key[lo], b = sm.key[hi];
                    if ((a > b) == up) { sm.key[lo] = b; sm.key[hi] = a; }
                }
                __syncthreads();
            }
        }
        if (wave0) {
            // element e of the walk order, e in [0, len]: e == len is position b1, which still holds its own unsorted
            // entry (the reference's inclusive upper bound, :151 / :190)
            auto elem = [&](const int e, float& r, int& q) {
                if (e < len) { const unsigned long long k = sm.key[e]; r = __uint_as_float((unsigned)(k >> 32)); q = (int)(unsigned)(k & 0xffffffffull); }
                else { r = sm.rough[b1]; q = b1; }
            };
            // corner loop: from the largest roughness down.  A chunk of 64 walk elements sits one per lane; the ballot of
            // "roughness above the threshold and still valid" is the work list, visited in order by bit scan, and every
            // suppression patches the prefetched flags of the lanes it hits before the list is re-read: the sequential
            // rule exactly, at the price of one loop trip per ACCEPTED point instead of one per element.
            int large = 0;
            bool stop = false;
            for (int c0 = len; c0 >= 0 && !stop; c0 -= 64) {  // chunk = walk elements c0, c0-1, ..., c0-63
                float rv = 0.f; int qv = 0, vv = 0;
                const int e_l = c0 - lane;
                const bool have = e_l >= 0;
                if (have) { elem(e_l, rv, qv); vv = sm.valid[qv]; }
                const bool cand = have && rv > p.corner_thr;
                unsigned long long m = __ballot(cand && vv);
                while (m) {
                    const int u = __builtin_ctzll(m);
                    ++large;
                    if (large > 20) { stop = true; break; }
                    const int q = __builtin_amdgcn_readlane(qv, u);
                    sm.corner[q] = 1;
                    if (lane == 0) crow[nc] = base + q;
                    ++nc;
                    int lo, hi;
                    suppress(q, lo, hi);
                    if (qv >= lo && qv <= hi) vv = 0;
                    m = __ballot(cand && vv) & (u == 63 ? 0ull : (~0ull << (u + 1)));
                }
                // a sorted element at or below the threshold ends the search (the boundary element e == len is not sorted)
                if (__ballot(have && !cand && e_l != len)) stop = true;
            }
            // planar loop: ascending; same scheme with "roughness below planar_thr and still valid" as the work list;
            // the emission of every non-corner element is order-preserving ballot compaction
            for (int c0 = 0; c0 <= len; c0 += 64) {
                float rv = 0.f; int qv = 0, vv = 0;
                const int e_l = c0 + lane;
                const bool have = e_l <= len;
                if (have) { elem(e_l, rv, qv); vv = sm.valid[qv]; }
                const bool cand = have && rv < p.planar_thr;
                unsigned long long m = __ballot(cand && vv);
                while (m) {
                    const int u = __builtin_ctzll(m);
                    int lo, hi;
                    suppress(__builtin_amdgcn_readlane(qv, u), lo, hi);
                    if (qv >= lo && qv <= hi) vv = 0;
                    m = __ballot(cand && vv) & (u == 63 ? 0ull : (~0ull << (u + 1)));
                }
                const bool emit = have && !sm.corner[qv];
                const unsigned long long me = __ballot(emit);
                if (emit) prow[np + __popcll(me & ((1ull << lane) - 1ull))] = base + qv;
                np += __popcll(me);
            }
        }
        __syncthreads();
    }
    if (tid == 0) { corner_cnt[row] = nc; planar_cnt[row] = np; }
    for (int k = tid; k < count; k += kFeatSelectThreads) {
        valid[base + k] = sm.valid[k];
        is_corner[base + k] = sm.corner[k];
    }
}

}  // namespace fls
