// host_maps.hpp -- host-side map bookkeeping of the product (C++), and flattening into the
// device image (hash table + voxel-bucketed float4 points) the kernels read.
//
//   HostIvox      : the iVox state with the reference's exact insert / LRU-evict semantics
//                   (src/ivox_map/ivox_map.cpp:122-143, capacity rule :133-136): the exact host path and
//                   the mirror of the device image (ivox_image.hpp; kernels_ivox_update.hpp maintains it).
//   voxel_grid    : pcl::VoxelGrid<PointXYZI>::filter semantics (centroid per leaf, ascending
//                   leaf index) used inside Match by ICP/NDT and by the kd-tree map updates
//                   (include/common/pointcloud_utility.h:216-271).
//   CellGridImage : exact-kNN uniform grid over a flat map cloud (stand-in for the reference's
//                   pcl::KdTreeFLANN, see kernels_knn.hpp for the exactness argument).
#pragma once
#include "device_common.hpp"
#include "host_util.hpp"
#include "host_parallel.hpp"
#include <unordered_map>
#include <algorithm>
#include <climits>
#include <cmath>
#include <limits>

namespace fls {

struct Pt4 { float x, y, z; int id; };  // same 16 bytes as the device float4 {x,y,z,id-bits}
static_assert(sizeof(Pt4) == 16, "Pt4 must alias float4");

struct PtI { float x, y, z, i; };  // xyz + intensity (VoxelGrid averages all fields)

// intensity of one AoS point: pcl::PointXYZI (stride 8) keeps it in float 4 (x, y, z, pad | intensity, pad x3),
// a packed xyzi row (stride 4..7) in float 3
inline float intensity_of(const float* q, int stride) { return stride >= 8 ? q[4] : stride >= 4 ? q[3] : 0.0f; }

inline std::vector<PtI> cloud_from(const float* p, size_t n, int stride) {
    std::vector<PtI> c(n);
    for (size_t k = 0; k < n; ++k) {
        c[k].x = p[k * stride];
        c[k].y = p[k * stride + 1];
        c[k].z = p[k * stride + 2];
        c[k].i = intensity_of(p + k * stride, stride);
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// voxel key -> pool slot: open addressing, linear probing, backward-shift deletion (no tombstones).  The insert
// path of the map is one probe sequence in one array instead of a bucket + node chase of std::unordered_map
// (AddPoints: 150 ns -> 40 ns per point at 180k voxels).
class FlatKeyMap {
public:
    int find(unsigned long long key) const {
        if (cap_ == 0) return -1;
        for (size_t h = hash_key(key) & mask_;; h = (h + 1) & mask_) {
            if (keys_[h] == key) return vals_[h];
            if (keys_[h] == kEmptyKey) return -1;
        }
    }
    void insert(unsigned long long key, int val) {  // key must be absent
        if ((size_ + 1) * 2 > cap_) grow();
        size_t h = hash_key(key) & mask_;
        while (keys_[h] != kEmptyKey) h = (h + 1) & mask_;
        keys_[h] = key;
        vals_[h] = val;
        ++size_;
    }
    void erase(unsigned long long key) {
        if (cap_ == 0) return;
        size_t h = hash_key(key) & mask_;
        while (keys_[h] != key) {
            if (keys_[h] == kEmptyKey) return;
            h = (h + 1) & mask_;
        }
        // backward shift: pull later members of the probe run into the hole when that keeps them reachable
        size_t hole = h;
        for (size_t j = (h + 1) & mask_; keys_[j] != kEmptyKey; j = (j + 1) & mask_) {
            const size_t home = hash_key(keys_[j]) & mask_;
            const bool reachable_from_hole = ((j - home) & mask_) >= ((j - hole) & mask_);
            if (reachable_from_hole) { keys_[hole] = keys_[j]; vals_[hole] = vals_[j]; hole = j; }
        }
        keys_[hole] = kEmptyKey;
        --size_;
    }
    void clear() { keys_.clear(); vals_.clear(); cap_ = mask_ = size_ = 0; }
    size_t size() const { return size_; }

private:
    void grow() {
        const size_t nc = cap_ ? cap_ * 2 : 1024;
        std::vector<unsigned long long> ok;
        std::vector<int> ov;
        ok.swap(keys_); ov.swap(vals_);
        keys_.assign(nc, kEmptyKey);
        vals_.assign(nc, -1);
        cap_ = nc; mask_ = nc - 1; size_ = 0;
        for (size_t i = 0; i < ok.size(); ++i)
            if (ok[i] != kEmptyKey) insert(ok[i], ov[i]);
    }
    std::vector<unsigned long long> keys_;
    std::vector<int> vals_;
    size_t cap_ = 0, mask_ = 0, size_ = 0;
};

class HostIvox {
public:
    struct Voxel {
        unsigned long long key;
        std::vector<Pt4> pts;
        int prev = -1, next = -1;  // LRU list (head = most recently inserted-into)
        bool alive = false;
        // device-image bookkeeping (GridImage): slot region [img_begin, img_begin + img_cap), img_cnt points mirrored
        unsigned img_begin = 0, img_cap = 0, img_cnt = 0;
        bool dirty = false;
    };
    // journal of the changes since the device image was last synchronised
    std::vector<int> touched;                       // voxel slots with new points (deduplicated through Voxel::dirty)
    std::vector<unsigned long long> evicted_keys;   // voxels dropped by the LRU rule
    void clear_journal() {
        for (int v : touched) pool[v].dirty = false;
        touched.clear();
        evicted_keys.clear();
    }
    float resolution = 0.5f, inv_resolution = 2.0f;
    size_t capacity = 1000000;
    std::vector<Voxel> pool;
    std::vector<int> free_slots;
    FlatKeyMap index;
    int head = -1, tail = -1;
    size_t n_alive = 0, n_points = 0;
    int next_id = 0;

    void clear() {
        pool.clear(); free_slots.clear(); index.clear(); touched.clear(); evicted_keys.clear();
        head = tail = -1; n_alive = n_points = 0; next_id = 0;
    }
    static bool key_of(float x, float y, float z, float inv, int& kx, int& ky, int& kz) {
        const float fx = std::round(x * inv), fy = std::round(y * inv), fz = std::round(z * inv);
        if (!(std::fabs(fx) < float(kKeyLimit) && std::fabs(fy) < float(kKeyLimit) && std::fabs(fz) < float(kKeyLimit))) return false;
        kx = int(fx); ky = int(fy); kz = int(fz);
        return true;
    }
    // all-or-nothing range check, then the reference's sequential insert
    fls_status add_points(const PtI* pts, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            int a, b, c;
            if (!key_of(pts[i].x, pts[i].y, pts[i].z, inv_resolution, a, b, c)) return FLS_ERR_RANGE;
        }
        unsigned long long last_key = kEmptyKey;  // consecutive points mostly share a voxel: that voxel is already at the
        int last_v = -1;                          // LRU front, so neither the lookup nor the splice is needed
        for (size_t i = 0; i < n; ++i) {
            int kx, ky, kz;
            key_of(pts[i].x, pts[i].y, pts[i].z, inv_resolution, kx, ky, kz);
            const unsigned long long key = pack_key(kx, ky, kz);
            const Pt4 p{pts[i].x, pts[i].y, pts[i].z, next_id++};
            if (key == last_key) {
                pool[last_v].pts.push_back(p);
                ++n_points;
                continue;
            }
            int v = index.find(key);
            if (v < 0) {
                if (!free_slots.empty()) { v = free_slots.back(); free_slots.pop_back(); }
                else { v = int(pool.size()); pool.emplace_back(); }
                Voxel& vx = pool[v];
                vx.key = key; vx.pts.clear(); vx.pts.push_back(p); vx.alive = true;
                vx.img_begin = vx.img_cap = vx.img_cnt = 0;
                if (!vx.dirty) { vx.dirty = true; touched.push_back(v); }
                link_front(v);
                index.insert(key, v);
                ++n_alive; ++n_points;
                if (n_alive >= capacity) evict_tail();
                // (capacity >= 2: the voxel just created is at the front, never the one evicted)
            } else {
                pool[v].pts.push_back(p);
                if (!pool[v].dirty) { pool[v].dirty = true; touched.push_back(v); }
                ++n_points;
                if (head != v) { unlink(v); link_front(v); }
            }
            last_key = key;
            last_v = v;
        }
        return FLS_OK;
    }

    // Rebuild the mirror from a downloaded device image (the device was authoritative: kernels_ivox_update.hpp).  Voxels are
    // given in any order with their LRU stamp; the list is re-linked so that larger stamps sit nearer the head.
    struct ImageVoxel { unsigned long long key; unsigned begin, count, cap; unsigned long long stamp; };  // 64-bit stamps: 2^32 inserted points is a day or two of mapping
    void rebuild_from_image(std::vector<ImageVoxel>& vox, const Pt4* pts, size_t n_points_total, int next_id_now) {
        clear();
        std::sort(vox.begin(), vox.end(), [](const ImageVoxel& a, const ImageVoxel& b) { return a.stamp < b.stamp; });
        pool.resize(vox.size());
        for (size_t i = 0; i < vox.size(); ++i) {  // ascending stamp: every link_front puts a more recent voxel in front
            Voxel& v = pool[i];
            v.key = vox[i].key;
            v.pts.assign(pts + vox[i].begin, pts + vox[i].begin + vox[i].count);
            v.alive = true; v.dirty = false;
            v.img_begin = vox[i].begin; v.img_cap = vox[i].cap; v.img_cnt = vox[i].count;
            link_front(int(i));
            index.insert(v.key, int(i));
        }
        n_alive = vox.size();
        n_points = n_points_total;
        next_id = next_id_now;
    }

private:
    void link_front(int v) {
        pool[v].prev = -1; pool[v].next = head;
        if (head >= 0) pool[head].prev = v;
        head = v;
        if (tail < 0) tail = v;
    }
    void unlink(int v) {
        const int p = pool[v].prev, nx = pool[v].next;
        if (p >= 0) pool[p].next = nx; else head = nx;
        if (nx >= 0) pool[nx].prev = p; else tail = p;
    }
    void evict_tail() {
        const int v = tail;
        if (v < 0) return;
        unlink(v);
        index.erase(pool[v].key);
        evicted_keys.push_back(pool[v].key);
        n_points -= pool[v].pts.size();
        pool[v].pts.clear(); pool[v].pts.shrink_to_fit();
        pool[v].alive = false;
        free_slots.push_back(v);
        --n_alive;
    }
};

// ---------------------------------------------------------------------------------------------
// Device image of a cell grid (the kd-tree kinds' CellGridImage): hash table and / or dense cell window.
// (The iVox map has its own two-level brick image: ivox_image.hpp.)
// ---------------------------------------------------------------------------------------------
struct GridImage {
    std::vector<HashEntry> table;
    std::vector<Pt4> pts;
    DevBuf<HashEntry> d_table;
    DevBuf<float4> d_pts;
    unsigned mask = 0;
    size_t used = 0;  // slots in use on the device (iVox images: includes per-voxel slack); 0 for kd-kind grids

    static unsigned table_size_for(size_t n_keys) {
        size_t s = 1024;
        while (s < 2 * n_keys + 2) s <<= 1;
        return unsigned(s);
    }
    void begin_build(size_t n_keys, size_t n_pts) {
        const unsigned ts = table_size_for(n_keys);
        mask = ts - 1;
        table.assign(ts, HashEntry{kEmptyKey, 0u, 0u});
        pts.clear();
        pts.reserve(n_pts);
    }
    void insert_bucket(unsigned long long key, const Pt4* p, size_t n) {
        unsigned h = hash_key(key) & mask;
        while (table[h].key != kEmptyKey) h = (h + 1) & mask;
        table[h] = HashEntry{key, unsigned(pts.size()), unsigned(n)};
        pts.insert(pts.end(), p, p + n);
    }
    void upload(hipStream_t s) {
        d_table.reserve(table.size());
        d_pts.reserve(std::max<size_t>(pts.size(), 1));
        FLS_HIP(hipMemcpyAsync(d_table.p, table.data(), table.size() * sizeof(HashEntry), hipMemcpyHostToDevice, s));
        if (!pts.empty()) FLS_HIP(hipMemcpyAsync(d_pts.p, pts.data(), pts.size() * sizeof(Pt4), hipMemcpyHostToDevice, s));
        FLS_HIP(hipStreamSynchronize(s));
    }
    DevGrid dev() const { return DevGrid{d_table.p, d_pts.p, mask, unsigned(used ? used : pts.size())}; }

    // dense window (see device_common.hpp DenseWindow)
    std::vector<uint2> cells;
    DevBuf<uint2> d_cells;
    int win_o[3] = {0, 0, 0}, win_n[3] = {0, 0, 0};
    bool have_window = false;
    static constexpr size_t kMaxWindowCells = size_t(48) << 20;  // 48 Mi cells = 384 MiB of {begin,count}
    DenseWindow window() const {
        return have_window ? DenseWindow{d_cells.p, win_o[0], win_o[1], win_o[2], win_n[0], win_n[1], win_n[2]}
                           : DenseWindow{nullptr, 0, 0, 0, 0, 0, 0};
    }

    bool cell_index(int x, int y, int z, size_t& idx) const {
        const long cx = long(x) - win_o[0], cy = long(y) - win_o[1], cz = long(z) - win_o[2];
        if (cx < 0 || cy < 0 || cz < 0 || cx >= win_n[0] || cy >= win_n[1] || cz >= win_n[2]) return false;
        idx = (size_t(cz) * win_n[1] + size_t(cy)) * win_n[0] + size_t(cx);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// pcl::VoxelGrid<PointXYZI>::filter (PCL 1.10 applyFilter; downsample_all_data_ = true)
// ---------------------------------------------------------------------------------------------
struct VoxelLeafRec { unsigned idx, pt; bool operator<(const VoxelLeafRec& o) const { return idx < o.idx; } };

inline std::vector<PtI> voxel_grid_sequential(const std::vector<PtI>& in, float leaf) {
    std::vector<PtI> out;
    if (in.empty()) return out;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const PtI& p : in) {
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
        mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
        mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
    }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                    dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (long long)std::numeric_limits<int>::max()) return in;  // PCL: "leaf size too small", input copied
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = int(std::floor(mn[a] * inv));
        div_b[a] = int(std::floor(mx[a] * inv)) - min_b[a] + 1;
    }
    const int m1 = div_b[0], m2 = div_b[0] * div_b[1];
    std::vector<VoxelLeafRec> lv;
    lv.reserve(in.size());
    for (size_t k = 0; k < in.size(); ++k) {
        const PtI& p = in[k];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        const int i0 = int(std::floor(p.x * inv) - float(min_b[0]));
        const int i1 = int(std::floor(p.y * inv) - float(min_b[1]));
        const int i2 = int(std::floor(p.z * inv) - float(min_b[2]));
        lv.push_back(VoxelLeafRec{unsigned(i0 + i1 * m1 + i2 * m2), unsigned(k)});
    }
    std::sort(lv.begin(), lv.end());
    for (size_t a = 0; a < lv.size();) {
        size_t b = a + 1;
        while (b < lv.size() && lv[b].idx == lv[a].idx) ++b;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t k = a; k < b; ++k) { const PtI& p = in[lv[k].pt]; sx += p.x; sy += p.y; sz += p.z; si += p.i; }
        const float cnt = float(b - a);
        out.push_back(PtI{sx / cnt, sy / cnt, sz / cnt, si / cnt});
        a = b;
    }
    return out;
}

// The same filter on the host worker pool (host_parallel.hpp): bounds, leaf indices and centroids are data-parallel chunks,
// the sort is the EXACT parallel restatement of std::sort (same permutation of equal leaf indices => the same float sums, bit
// for bit).  Falls back to the sequential code when the pool is busy / single-threaded or introsort would have heap-sorted.
// a caller's strided AoS cloud read in place (no intermediate copy): the same values cloud_from() would produce
struct StridedCloud {
    const float* p;
    size_t n;
    int stride;
    size_t size() const { return n; }
    PtI operator[](size_t k) const { const float* q = p + k * size_t(stride); return PtI{q[0], q[1], q[2], intensity_of(q, stride)}; }
};
inline void copy_cloud(const std::vector<PtI>& in, std::vector<PtI>& out) { out = in; }
inline void copy_cloud(const StridedCloud& in, std::vector<PtI>& out) { out = cloud_from(in.p, in.n, in.stride); }

template <class Cloud>
inline bool voxel_grid_parallel(const Cloud& in, float leaf, std::vector<PtI>& out) {
    HostPool& pool = HostPool::get();
    const size_t n = in.size();
    const float inv = 1.0f / leaf;
    const size_t C = size_t(pool.threads()) * 4;  // chunks per phase
    struct Chunk { float mn[3], mx[3]; size_t finite, offset; std::vector<PtI> part; };
    std::vector<Chunk> ch(C);
    std::vector<VoxelLeafRec> lv;
    ExactSortShared<VoxelLeafRec> sh;
    int verdict = 0;  // 1: input copied ("leaf size too small"), 2: sort needs the sequential path, 3: no finite point
    const bool ran = pool.run([&](HostPool::Region& reg) {
        auto finite = [](const PtI& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); };
        reg.phase(C, [&](const size_t c) {
            Chunk& me = ch[c];
            for (int k = 0; k < 3; ++k) { me.mn[k] = INFINITY; me.mx[k] = -INFINITY; }
            me.finite = 0;
            for (size_t k = n * c / C, e = n * (c + 1) / C; k < e; ++k) {
                const PtI p = in[k];
                if (!finite(p)) continue;
                ++me.finite;
                me.mn[0] = std::min(me.mn[0], p.x); me.mx[0] = std::max(me.mx[0], p.x);
                me.mn[1] = std::min(me.mn[1], p.y); me.mx[1] = std::max(me.mx[1], p.y);
                me.mn[2] = std::min(me.mn[2], p.z); me.mx[2] = std::max(me.mx[2], p.z);
            }
        });
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        size_t total = 0;
        for (Chunk& c : ch) {
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], c.mn[k]); mx[k] = std::max(mx[k], c.mx[k]); }
            c.offset = total;
            total += c.finite;
        }
        if (total == 0) { verdict = 3; return; }
        const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                        dz = (long long)((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > (long long)std::numeric_limits<int>::max()) { verdict = 1; return; }
        int min_b[3], div_b[3];
        for (int k = 0; k < 3; ++k) {
            min_b[k] = int(std::floor(mn[k] * inv));
            div_b[k] = int(std::floor(mx[k] * inv)) - min_b[k] + 1;
        }
        const int m1 = div_b[0], m2 = div_b[0] * div_b[1];
        lv.resize(total);
        reg.phase(C, [&](const size_t c) {
            size_t w = ch[c].offset;
            for (size_t k = n * c / C, e = n * (c + 1) / C; k < e; ++k) {
                const PtI p = in[k];
                if (!finite(p)) continue;
                const int i0 = int(std::floor(p.x * inv) - float(min_b[0]));
                const int i1 = int(std::floor(p.y * inv) - float(min_b[1]));
                const int i2 = int(std::floor(p.z * inv) - float(min_b[2]));
                lv[w++] = VoxelLeafRec{unsigned(i0 + i1 * m1 + i2 * m2), unsigned(k)};
            }
        });
        sh.reset(lv.data(), lv.data() + total);
        reg.phase(size_t(pool.threads()), [&](size_t) { exact_sort_worker(sh); });
        if (sh.failed.load()) { verdict = 2; return; }
        // centroids: chunk c owns the leaves that START in its slice of the sorted records
        reg.phase(C, [&](const size_t c) {
            size_t s0 = total * c / C, s1 = total * (c + 1) / C;
            while (s0 > 0 && s0 < total && lv[s0].idx == lv[s0 - 1].idx) ++s0;
            while (s1 > 0 && s1 < total && lv[s1].idx == lv[s1 - 1].idx) ++s1;
            std::vector<PtI>& part = ch[c].part;
            part.clear();
            for (size_t x = s0; x < s1;) {
                size_t y = x + 1;
                while (y < total && lv[y].idx == lv[x].idx) ++y;
                float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
                for (size_t k = x; k < y; ++k) { const PtI p = in[lv[k].pt]; sx += p.x; sy += p.y; sz += p.z; si += p.i; }
                const float cnt = float(y - x);
                part.push_back(PtI{sx / cnt, sy / cnt, sz / cnt, si / cnt});
                x = y;
            }
        });
    });
    if (!ran || verdict == 2) return false;   // the sequential path works on the untouched input
    if (verdict == 1) { copy_cloud(in, out); return true; }  // PCL: "leaf size too small", input copied
    out.clear();
    if (verdict == 3) return true;
    size_t tot = 0;
    for (const Chunk& c : ch) tot += c.part.size();
    out.reserve(tot);
    for (const Chunk& c : ch) out.insert(out.end(), c.part.begin(), c.part.end());
    return true;
}

inline std::vector<PtI> voxel_grid(const std::vector<PtI>& in, float leaf) {
    // the pool pays from ~50k points on (waking the workers costs 0.1-0.2 ms; 27k points: 0.8 ms alone, 1.05 ms pooled;
    // 115k points: 2.6 ms alone, 0.75 ms on 8 threads)
    // ... and from 16k points on while the workers are still spinning after a region that ended less than a millisecond ago
    // (the second VoxelGrid of an NDT Match follows the first within that window)
    if (in.size() >= 49152 || (in.size() >= 16384 && HostPool::get().hot())) {
        std::vector<PtI> out;
        if (voxel_grid_parallel(in, leaf, out)) return out;
    }
    return voxel_grid_sequential(in, leaf);
}
// the filter of a caller's strided cloud (Match's source filter): large clouds are read in place by the pooled filter
inline std::vector<PtI> voxel_grid_strided(const float* p, size_t n, int stride, float leaf) {
    if (n >= 49152) {
        std::vector<PtI> out;
        if (voxel_grid_parallel(StridedCloud{p, n, stride}, leaf, out)) return out;
    }
    return voxel_grid_sequential(cloud_from(p, n, stride), leaf);
}

// ---------------------------------------------------------------------------------------------
// Exact-kNN cell grid over a flat cloud: cell = floor(p * inv_cell), ids = index in the cloud.
// ---------------------------------------------------------------------------------------------
struct CellGridImage : GridImage {
    float cell = 1.0f, inv_cell = 1.0f;
    int rings = 1;  // 2: the cell is half the gate radius, the search covers the 5x5x5 block in two stages
    size_t n_cells = 0;
    DevBuf<float4> d_by_id;  // optional: the cloud in its own order (see CellGridDev::by_id)
    fls_status build(const std::vector<PtI>& cloud, float cell_size, hipStream_t s, int n_rings = 1, bool with_by_id = false) {
        rings = n_rings;
        used = 0;  // (a grid the device builder filled earlier left its own point count here: dev() must report THIS build's)
        if (with_by_id && !cloud.empty()) {
            std::vector<Pt4> ordered(cloud.size());
            for (size_t i = 0; i < cloud.size(); ++i) ordered[i] = Pt4{cloud[i].x, cloud[i].y, cloud[i].z, int(i)};
            d_by_id.reserve(cloud.size());
            FLS_HIP(hipMemcpyAsync(d_by_id.p, ordered.data(), ordered.size() * sizeof(Pt4), hipMemcpyHostToDevice, s));
            FLS_HIP(hipStreamSynchronize(s));
        }
        cell = cell_size;
        inv_cell = 1.0f / cell_size;
        const size_t n = cloud.size();
        std::vector<std::pair<unsigned long long, unsigned>> kv(n);
        for (size_t i = 0; i < n; ++i) {
            const float fx = std::floor(cloud[i].x * inv_cell), fy = std::floor(cloud[i].y * inv_cell), fz = std::floor(cloud[i].z * inv_cell);
            if (!(std::fabs(fx) < float(kKeyLimit) && std::fabs(fy) < float(kKeyLimit) && std::fabs(fz) < float(kKeyLimit))) return FLS_ERR_RANGE;
            kv[i] = {pack_key(int(fx), int(fy), int(fz)), unsigned(i)};
        }
        pooled_sort_total_order(kv.data(), kv.data() + n);  // (cell key, index) pairs: a total order
        size_t cells = 0;
        for (size_t a = 0; a < n; ++a) if (a == 0 || kv[a].first != kv[a - 1].first) ++cells;
        n_cells = cells;
        begin_build(cells, n);
        // dense cell window {begin, count} over the bounding box of the occupied cells (one aligned 8-byte load per
        // probe instead of a hash probe sequence; neighbouring cells share cache lines), when the box is small enough
        int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
        for (size_t a = 0; a < n; ++a) {
            if (a && kv[a].first == kv[a - 1].first) continue;
            int x, y, z;
            unpack_key(kv[a].first, x, y, z);
            mn[0] = std::min(mn[0], x); mx[0] = std::max(mx[0], x);
            mn[1] = std::min(mn[1], y); mx[1] = std::max(mx[1], y);
            mn[2] = std::min(mn[2], z); mx[2] = std::max(mx[2], z);
        }
        have_window = false;
        this->cells.clear();
        if (n) {
            size_t nc = 1;
            for (int a = 0; a < 3; ++a) { win_o[a] = mn[a]; win_n[a] = mx[a] - mn[a] + 1; nc *= size_t(win_n[a]); }
            have_window = nc <= kMaxWindowCells;
            if (have_window) this->cells.assign(nc, make_uint2(0u, 0u));
        }
        std::vector<Pt4> bucket;
        for (size_t a = 0; a < n;) {
            size_t b = a + 1;
            while (b < n && kv[b].first == kv[a].first) ++b;
            bucket.clear();
            for (size_t k = a; k < b; ++k) { const PtI& p = cloud[kv[k].second]; bucket.push_back(Pt4{p.x, p.y, p.z, int(kv[k].second)}); }
            if (have_window) {
                int x, y, z;
                unpack_key(kv[a].first, x, y, z);
                size_t idx;
                if (cell_index(x, y, z, idx)) this->cells[idx] = make_uint2(unsigned(pts.size()), unsigned(bucket.size()));
            }
            insert_bucket(kv[a].first, bucket.data(), bucket.size());
            a = b;
        }
        upload(s);
        if (have_window) {
            d_cells.reserve(this->cells.size());
            FLS_HIP(hipMemcpyAsync(d_cells.p, this->cells.data(), this->cells.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
            FLS_HIP(hipStreamSynchronize(s));
            std::vector<uint2>().swap(this->cells);  // the host copy is not needed again (the grid is rebuilt, never patched)
        }
        return FLS_OK;
    }
};

}  // namespace fls
