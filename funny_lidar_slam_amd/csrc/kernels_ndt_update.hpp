// kernels_ndt_update.hpp -- IncrementalNDT::AddCloud on the device (second half).
//
// Reference: incremental_ndt.h:182-227 (per point: key = cast<int>(p * inv_voxel_size); new voxel -> push_front + capacity
// check, known voxel -> append + splice to the front; then UpdateVoxel of every touched voxel, :130-179).  This file covers the
// mapping-mode steady state: flag_first_scan == false, with the LRU evictions a batch causes (ndt_evict_select).  Anything
// else (first scan, localization mode, evictions that would reach voxels of the batch itself, a key out of range) is refused without side effects and runs
// through the exact sequential host code (matcher_ndt.hpp), which first downloads the device state.
//
// How the sequential semantics are recovered from a parallel pass:
//   * a voxel's points are consumed in cloud order: the {row, point} pairs are sorted by row with the stable radix sort of
//     kernels_voxelgrid.hpp, so a segment lists its points in ascending cloud index -- the order of v.pts;
//   * voxel ids (the reference's creation order, visible as correspondence ids) follow the cloud index of the FIRST point of
//     every new voxel: an atomicMax per new table entry on (2^31 - 1 - index) erases the arrival order of the threads, an
//     exclusive scan over "this point creates a voxel" gives the creation rank;
//   * the LRU order (std::list splice per point) is kept as a stamp per voxel: epoch + cloud index of its last point;
//   * a voxel keeps at most ndt_min_points (5) unconsumed points between scans (more triggers an estimate, which clears
//     them; a saturated voxel's list is never read again): a fixed carry buffer per row;
//   * the statistics (mean / covariance, pooled update, 3x3 Jacobi SVD clamp, 3x3 inverse) are the host's own functions
//     compiled for the device (host_math.hpp, __host__ __device__): bit-identical rows.
#pragma once
#include "kernels_knn.hpp"
#include "kernels_voxelgrid.hpp"
#include "host_math.hpp"

namespace fls {

constexpr unsigned kNdtNewBit = 0x80000000u;  // HashEntry.begin of an EMPTY or freshly claimed entry: 2^31 | (2^31 - 1 - first index)
constexpr int kNdtCarry = 8;                  // unconsumed points kept per voxel between scans (min_points <= kNdtCarry)
constexpr unsigned kNdtOk = 0u, kNdtNeedHost = 1u;
constexpr unsigned long long kNdtTombKey = ~0ull - 1ull;  // table entry of an evicted voxel: probes walk past it (never equal to a packed key, not EMPTY)
constexpr unsigned long long kNdtDeadKey = ~0ull - 2ull;  // r.key of an evicted row

struct NdtRows {
    unsigned long long* key;   // packed voxel key
    unsigned* hslot;           // index of the voxel's table entry
    int* num_points;
    unsigned char* estimated;
    unsigned char* carry_cnt;
    double* carry;             // [row][kNdtCarry][3]
    double* mu;                // [row][3]   (read by ndt_kernel)
    double* sigma;             // [row][9]
    double* info;              // [row][9]   (read by ndt_kernel)
    int* vid;                  // [row]      (read by ndt_kernel)
    unsigned long long* stamp; // LRU: larger = touched later
    unsigned long long* touch; // {sequence number of the last batch that touched the row : 2^31 - 1 - cloud index of that batch's first point in it} (eviction selection)
};

struct NdtUpdState {
    unsigned n_rows, n_alive;
    int next_vid;
    unsigned pad0;
    unsigned long long epoch;
    unsigned capacity, row_cap;
    unsigned n_new, status, apply, touched;
    unsigned evict, seq, evict_ready, dead_hi;  // voxels this batch evicts; batch sequence number; the host queued the selection; hi word given to dead rows' sort key
    unsigned recreated, pad1;                   // voxels this batch evicts AND re-creates (touched after their turn in the eviction order: ndt_evict_select)
};
constexpr unsigned kNdtRecreateBit = 0x80000000u;  // evict_list entry: the row is retired but its table entry lives on (the re-created voxel's row takes it over)
constexpr int kNdtEvBlock = 1024, kNdtMaxRecreate = 1024;
__device__ __forceinline__ unsigned long long ndt_touch_word(const unsigned seq, const unsigned i) { return ((unsigned long long)seq << 32) | (unsigned long long)(0x7fffffffu - i); }
// number of entries of the ascending list v[0, n) that are smaller than x
__device__ __forceinline__ unsigned ndt_lower_bound(const unsigned* __restrict__ v, const unsigned n, const unsigned x) {
    unsigned lo = 0u, hi = n;
    while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (v[mid] < x) lo = mid + 1u; else hi = mid; }
    return lo;
}

// per point: voxel key, find or claim its table entry; for a claimed (new) entry remember the smallest cloud index
__global__ void __launch_bounds__(256)
ndt_upd_locate(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const double inv_voxel,
               HashEntry* __restrict__ table, const unsigned mask, unsigned* __restrict__ slot_h, NdtUpdState* __restrict__ st, unsigned long long* __restrict__ touch,
               const unsigned seq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double f0 = (double)x[i] * inv_voxel, f1 = (double)y[i] * inv_voxel, f2 = (double)z[i] * inv_voxel;
    if (!(fabs(f0) < (double)kKeyLimit) || !(fabs(f1) < (double)kKeyLimit) || !(fabs(f2) < (double)kKeyLimit)) {
        atomicOr(&st->status, kNdtNeedHost);  // FLS_ERR_RANGE on the host path: all-or-nothing
        slot_h[i] = 0xffffffffu;
        return;
    }
    const unsigned long long key = pack_key((int)f0, (int)f1, (int)f2);  // cast<int>: truncation (:195)
    unsigned h = hash_key(key) & mask;
    for (;;) {
        unsigned long long k = __hip_atomic_load(&table[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == kEmptyKey) {
            const unsigned long long old = atomicCAS(&table[h].key, kEmptyKey, key);
            k = old == kEmptyKey ? key : old;
        }
        if (k == key) break;
        h = (h + 1) & mask;
    }
    slot_h[i] = h;
    const unsigned b = __hip_atomic_load(&table[h].begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b >= kNdtNewBit) atomicMax(&table[h].begin, kNdtNewBit | (0x7fffffffu - (unsigned)i));
    else atomicMax(&touch[b], ndt_touch_word(seq, (unsigned)i));  // an existing voxel this batch appends to: a newer batch beats an older one, the smallest index wins inside one
}

// creators (first point of a new voxel): block-local exclusive rank + block totals
__global__ void __launch_bounds__(kVgScanBlock)
ndt_upd_creators(const int n, const HashEntry* __restrict__ table, const unsigned* __restrict__ slot_h, unsigned* __restrict__ lx,
                 unsigned* __restrict__ bt) {
    __shared__ unsigned wsum[kVgScanBlock / 64];
    const int i = blockIdx.x * kVgScanBlock + threadIdx.x;
    unsigned c = 0u;
    if (i < n && slot_h[i] != 0xffffffffu) {
        const unsigned b = table[slot_h[i]].begin;
        c = (b >= kNdtNewBit && (0x7fffffffu - (b & 0x7fffffffu)) == (unsigned)i) ? 1u : 0u;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0u, tot = 0u;
    for (int q = 0; q < kVgScanBlock / 64; ++q) { const unsigned t = wsum[q]; if (q < w) base += t; tot += t; }
    if (i < n) lx[i] = c ? (base + inc - 1u) : 0xffffffffu;
    if (threadIdx.x == 0) bt[blockIdx.x] = tot;
}

// LRU evictions inside a batch (incremental_ndt.h:202-206: a creation that brings the count to the capacity pops the list's back).
// With n alive voxels and k creations the batch evicts E = max(0, n + k - (capacity - 1)) voxels: the tail at each eviction; eviction
// number idx happens right after creation number (capacity - 1 - n) + idx.  The candidates are the rows in LRU order (the host sorts
// them by stamp -- two stable 32-bit radix rounds, dead rows last -- whenever the batch COULD reach the capacity).  An untouched
// candidate is the next eviction; one the batch touches BEFORE that eviction's creation has moved to the front and is skipped; one it
// touches AFTER it is evicted and RE-CREATED by its first point (one more creation -- its index joins the creation sequence -- one more
// eviction, a new voxel id, statistics from the batch's points alone).  Round 3 refused every batch that touched one of the E oldest
// rows; round 4 walks the list exactly like the iVox kind does (ndt_evict_select below = ivox_evict_select, kernels_ivox_update.hpp;
// the walk is checked against the sequential loop in tests/host/evict_conflict_model_test.cpp).  What still goes to the host: a
// selection that would reach voxels this batch itself created or moved to the front (kNdtNeedHost).
__global__ void ndt_upd_predecide(NdtUpdState* __restrict__ st) {
    unsigned status = st->status;
    const unsigned long long total = (unsigned long long)st->n_alive + st->n_new;
    unsigned e = 0u;
    if (total >= (unsigned long long)st->capacity) {
        e = (unsigned)(total - (unsigned long long)st->capacity + 1ull);
        if (!st->evict_ready || e > st->n_alive) status |= kNdtNeedHost;
    }
    if ((unsigned long long)st->n_rows + st->n_new > (unsigned long long)st->row_cap) status |= kNdtNeedHost;
    st->evict = e;
    st->status = status;
}
// sort keys of the eviction selection: the stamp of every row, one 32-bit half per round (dead rows sort last)
__global__ void __launch_bounds__(256)
ndt_evict_keys(const NdtRows r, const unsigned n_rows, const unsigned dead_hi, const int round, const unsigned* __restrict__ order, unsigned* __restrict__ key,
               unsigned* __restrict__ val) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const unsigned row = round == 0 ? i : order[i];
    const bool dead = r.key[row] == kNdtDeadKey;
    const unsigned long long s = r.stamp[row];
    key[i] = round == 0 ? (dead ? 0xffffffffu : (unsigned)s) : (dead ? dead_hi : (unsigned)(s >> 32));
    if (round == 0) val[i] = row;
}
// cloud index of every creator, in creation (= cloud) order
__global__ void __launch_bounds__(256)
ndt_upd_cranks(const int n, const unsigned* __restrict__ lx, const unsigned* __restrict__ bt /* scanned */, unsigned* __restrict__ crank) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned l = lx[i];
    if (l != 0xffffffffu) crank[bt[i / kVgScanBlock] + l] = (unsigned)i;
}
// the eviction walk (one workgroup): evict_list[0, E + nS) = rows in eviction order (kNdtRecreateBit: re-created), s_idx / s_row = first
// cloud index (ascending) and old row of the re-created voxels; state: evict, recreated, n_new grow by nS
__global__ void __launch_bounds__(kNdtEvBlock)
ndt_evict_select(const NdtRows r, const unsigned* __restrict__ order, const unsigned n_rows, NdtUpdState* __restrict__ st, const unsigned* __restrict__ crank,
                 unsigned* __restrict__ evict_list, unsigned* __restrict__ s_idx_out, unsigned* __restrict__ s_row_out) {
    __shared__ unsigned wsum[kNdtEvBlock / 64];
    __shared__ unsigned s_found, s_nS, s_conflict, s_overflow;
    __shared__ unsigned s_S[kNdtMaxRecreate], s_R[kNdtMaxRecreate];
    const unsigned E = st->evict;
    if (E == 0u || st->status != kNdtOk) return;
    const unsigned C = st->n_new, seq = st->seq;
    const unsigned base_c = st->capacity - 1u > st->n_alive ? st->capacity - 1u - st->n_alive : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_found = 0u; s_nS = 0u; s_conflict = 0xFFFFFFFFu; s_overflow = 0u; }
    __syncthreads();
    for (unsigned j0 = 0; j0 < n_rows; j0 += kNdtEvBlock) {
        if (s_found >= E + s_nS) break;
        const unsigned j = j0 + threadIdx.x;
        unsigned row = 0u;
        bool valid = j < n_rows;
        if (valid) { row = order[j]; valid = r.key[row] != kNdtDeadKey; }  // (dead rows sort last: nothing behind the first one is a candidate)
        const unsigned long long tw = valid ? r.touch[row] : 0ull;
        const bool un = valid && (unsigned)(tw >> 32) != seq;
        const unsigned rv = 0x7fffffffu - (unsigned)(tw & 0x7fffffffull);  // first cloud index of a touched candidate
        unsigned start = j0;
        for (;;) {
            const unsigned found = s_found, nS = s_nS, Etot = E + nS;
            if (found >= Etot) break;
            const bool in = valid && j >= start;
            const unsigned long long m = __ballot(in && un);
            if (lane == 0) wsum[w] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned before = 0u, total = 0u;
            for (int q = 0; q < kNdtEvBlock / 64; ++q) { const unsigned t = wsum[q]; if (q < w) before += t; total += t; }
            const unsigned idx = found + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (in && !un && idx < Etot && !(rv < evict_merged_rank(crank, C, s_S, nS, base_c + idx))) atomicMin(&s_conflict, j);
            __syncthreads();
            const unsigned fc = s_conflict;
            if (in && un && j < fc && idx < Etot) evict_list[idx] = row;
            __syncthreads();
            if (fc == 0xFFFFFFFFu) {
                if (threadIdx.x == 0) s_found = found + total;
                __syncthreads();
                break;
            }
            if (j == fc) {
                evict_list[idx] = row | kNdtRecreateBit;
                if (nS >= (unsigned)kNdtMaxRecreate) {
                    s_overflow = 1u;
                } else {
                    unsigned q = nS;
                    while (q > 0u && s_S[q - 1] > rv) { s_S[q] = s_S[q - 1]; s_R[q] = s_R[q - 1]; --q; }
                    s_S[q] = rv; s_R[q] = row;
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
    __syncthreads();
    const unsigned nS = s_nS;
    if (s_overflow || s_found < E + nS) {
        if (threadIdx.x == 0) atomicOr(&st->status, kNdtNeedHost);
        return;
    }
    for (unsigned q = threadIdx.x; q < nS; q += kNdtEvBlock) { s_idx_out[q] = s_S[q]; s_row_out[q] = s_R[q]; }
    if (threadIdx.x == 0) { st->evict = E + nS; st->recreated = nS; st->n_new = C + nS; }
}
__global__ void ndt_upd_decide(NdtUpdState* __restrict__ st) { st->apply = st->status == kNdtOk ? 1u : 0u; }
// evict: tombstone the table entry, retire the row
__global__ void __launch_bounds__(256)
ndt_evict_apply(const NdtRows r, const unsigned* __restrict__ evict_list, HashEntry* __restrict__ table, const NdtUpdState* __restrict__ st) {
    const unsigned j = blockIdx.x * 256 + threadIdx.x;
    if (!st->apply || j >= st->evict) return;
    const unsigned ent = evict_list[j], row = ent & ~kNdtRecreateBit;
    const unsigned h = r.hslot[row];
    if (!(ent & kNdtRecreateBit)) table[h].key = kNdtTombKey;  // (a re-created voxel keeps its entry: ndt_upd_create points it at the new row)
    table[h].count = 0u;
    r.key[row] = kNdtDeadKey;
    r.estimated[row] = 0;
    r.carry_cnt[row] = 0;
}

// creators initialise their voxel's row (applied) or give the claimed entry back (refused)
__global__ void __launch_bounds__(256)
ndt_upd_create(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const double inv_voxel,
               HashEntry* __restrict__ table, const unsigned* __restrict__ slot_h, const unsigned* __restrict__ lx, const unsigned* __restrict__ bt,
               const NdtRows r, const NdtUpdState* __restrict__ st, const unsigned* __restrict__ crank, const unsigned* __restrict__ s_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned l = lx[i];
    // voxels the batch evicted before this point arrived are re-created by it (ndt_evict_select): their first points are creators too, and
    // every creation after one of them moves up by one (voxel ids and rows follow the MERGED creation order)
    const unsigned nS = st->apply ? st->recreated : 0u;
    unsigned sk = 0u;
    bool again = false;
    if (nS) { sk = ndt_lower_bound(s_idx, nS, (unsigned)i); again = sk < nS && s_idx[sk] == (unsigned)i; }
    if (l == 0xffffffffu && !again) return;
    const unsigned h = slot_h[i];
    if (!st->apply) {
        table[h] = HashEntry{kEmptyKey, kNdtNewBit, 0u};
        return;
    }
    const unsigned rank = again ? ndt_lower_bound(crank, st->n_new - nS, (unsigned)i) + sk : bt[i / kVgScanBlock] + l + sk;
    const unsigned row = st->n_rows + rank;
    const double f0 = (double)x[i] * inv_voxel, f1 = (double)y[i] * inv_voxel, f2 = (double)z[i] * inv_voxel;
    r.key[row] = pack_key((int)f0, (int)f1, (int)f2);
    r.hslot[row] = h;
    r.num_points[row] = 0;
    r.estimated[row] = 0;
    r.carry_cnt[row] = 0;
    r.vid[row] = st->next_vid + (int)rank;
    r.stamp[row] = 0ull;
    r.touch[row] = ndt_touch_word(st->seq, (unsigned)i);
    table[h].begin = row;
    table[h].count = 0u;  // not estimated yet: ndt_kernel skips it
}

// sort keys: the row of every point (a refused batch sorts zeros: nothing downstream looks at them)
__global__ void __launch_bounds__(256)
ndt_upd_rowkeys(const int n, const HashEntry* __restrict__ table, const unsigned* __restrict__ slot_h, unsigned* __restrict__ key, unsigned* __restrict__ val,
                const NdtUpdState* __restrict__ st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    key[i] = st->apply ? table[slot_h[i]].begin : 0u;
    val[i] = (unsigned)i;
}

// TransformPointCloud(cloud, Mat4d) (pointcloud_utility.h:141-195 == host_math.hpp::xform_cloud_f): R, t -> float, then
// float r0 x + (r1 y + r2 z) + t
struct XformF { float R[9], t[3]; };
__global__ void __launch_bounds__(256)
xform_cloud_f_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const XformF f,
                     float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = x[i], py = y[i], pz = z[i];
    ox[i] = (f.R[0] * px + (f.R[3] * py + f.R[6] * pz)) + f.t[0];
    oy[i] = (f.R[1] * px + (f.R[4] * py + f.R[7] * pz)) + f.t[1];
    oz[i] = (f.R[2] * px + (f.R[5] * py + f.R[8] * pz)) + f.t[2];
}

// device-side table rebuild (the table of a device-mode image grows without the host mirror)
__global__ void __launch_bounds__(256)
ndt_table_fill_kernel(HashEntry* __restrict__ table, const unsigned ts) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < ts) table[i] = HashEntry{kEmptyKey, kNdtNewBit, 0u};
}
__global__ void __launch_bounds__(256)
ndt_table_rehash_kernel(HashEntry* __restrict__ table, const unsigned mask, const NdtRows r, const unsigned n_rows) {
    const unsigned row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const unsigned long long key = r.key[row];
    if (key == kNdtDeadKey) return;  // evicted
    unsigned h = hash_key(key) & mask;
    while (atomicCAS(&table[h].key, kEmptyKey, key) != kEmptyKey) h = (h + 1) & mask;  // keys are distinct
    table[h].begin = row;
    table[h].count = r.estimated[row] ? 1u : 0u;
    r.hslot[row] = h;
}

// one thread per touched voxel (segment head of the sorted pairs): UpdateVoxel, incremental_ndt.h:130-179, flag_first_scan == false
__global__ void __launch_bounds__(64)
ndt_upd_voxel(const unsigned* __restrict__ key, const unsigned* __restrict__ val, const int n, const float* __restrict__ x,
              const float* __restrict__ y, const float* __restrict__ z, const NdtRows r, HashEntry* __restrict__ table,
              NdtUpdState* __restrict__ st, const int min_points, const int max_points) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n || !st->apply) return;
    const unsigned row = key[e];
    if (e > 0 && key[e - 1] == row) return;
    int len = 1;
    while (e + len < n && key[e + len] == row) ++len;
    atomicAdd(&st->touched, 1u);
    r.stamp[row] = st->epoch + (unsigned long long)val[e + len - 1] + 1ull;  // splice to the front at every point: the last one decides
    const bool est = r.estimated[row] != 0;
    int np = r.num_points[row];
    if (!est) { np += len; r.num_points[row] = np; }  // "if (!estimated) num_points++" per appended point (:207-209, 1 at creation)
    if (est && np > max_points) return;                // :140 (the list of a saturated voxel is never read again)
    const int cc = r.carry_cnt[row];
    const int npts = cc + len;
    double* const cb = r.carry + (size_t)row * (kNdtCarry * 3);
    if (npts <= min_points) {  // neither branch of :146 / :152 fires: the points wait
        for (int k = 0; k < len; ++k) {
            const unsigned p = val[e + k];
            cb[3 * (cc + k)] = (double)x[p]; cb[3 * (cc + k) + 1] = (double)y[p]; cb[3 * (cc + k) + 2] = (double)z[p];
        }
        r.carry_cnt[row] = (unsigned char)npts;
        return;
    }
    auto get = [=](const int k, double* q) {
        if (k < cc) { q[0] = cb[3 * k]; q[1] = cb[3 * k + 1]; q[2] = cb[3 * k + 2]; }
        else { const unsigned p = val[e + (k - cc)]; q[0] = (double)x[p]; q[1] = (double)y[p]; q[2] = (double)z[p]; }
    };
    double* const mu = r.mu + 3 * (size_t)row;
    double* const sigma = r.sigma + 9 * (size_t)row;
    double* const info = r.info + 9 * (size_t)row;
    if (!est) {
        double m[3], s[9], inf[9];
        hm::ndt_mean_cov(npts, get, m, s);
        hm::ndt_regularised_info(s, inf);
        for (int a = 0; a < 3; ++a) mu[a] = m[a];
        for (int a = 0; a < 9; ++a) { sigma[a] = s[a]; info[a] = inf[a]; }
        r.estimated[row] = 1;
        table[r.hslot[row]].count = 1u;  // visible to ndt_kernel from the next Match on
    } else {
        double cmu[3], cvar[9], m[3], s[9], inf[9];
        hm::ndt_mean_cov(npts, get, cmu, cvar);
        for (int a = 0; a < 3; ++a) m[a] = mu[a];
        for (int a = 0; a < 9; ++a) s[a] = sigma[a];
        hm::ndt_merge(m, s, inf, np, cmu, cvar, npts);
        for (int a = 0; a < 3; ++a) mu[a] = m[a];
        for (int a = 0; a < 9; ++a) { sigma[a] = s[a]; info[a] = inf[a]; }
        r.num_points[row] = np + npts;
    }
    r.carry_cnt[row] = 0;
}

__global__ void ndt_upd_commit(NdtUpdState* __restrict__ st, const unsigned n) {
    if (st->apply) {
        st->n_rows += st->n_new;
        st->n_alive += st->n_new;
        st->n_alive -= st->evict;
        st->next_vid += (int)st->n_new;
        st->epoch += (unsigned long long)n + 1ull;
    }
}

}  // namespace fls
