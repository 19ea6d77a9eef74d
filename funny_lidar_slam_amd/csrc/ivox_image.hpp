// ivox_image.hpp -- the device image of the iVox map (FLS_P2PLANE_IVOX) and its host-side bookkeeping.
//
//   points    float4 {x, y, z, insertion id}, bucketed by voxel; every voxel owns a power-of-two slot region (>= 4) so that
//             AddPoints appends in place (device_common.hpp / DESIGN.md 3)
//   voxels    two-level BRICK image (device_common.hpp BrickDir): directory {brick key -> slab}, slabs of 10^3 {begin, count} cells
//             (8^3 interior + a halo that mirrors the neighbouring bricks' boundary cells).  No extent limit: the reference's IVoxMap
//             bounds the number of voxels only (include/ivox_map/ivox_map.h:35, src/ivox_map/ivox_map.cpp:122-147).
//             FLS_IVOX_DENSE=0 (A/B): a per-voxel open-addressing hash table instead (host-maintained, no device AddPoints).
//
// The image is built from the host mirror (HostIvox) by uploading the points and per-voxel RECORDS {cell, begin, count} that a scatter
// kernel writes into the zeroed slabs -- a few MB for a 1e6-point map instead of the slabs themselves -- and is then maintained either by
// the device-side AddPoints (kernels_ivox_update.hpp; the device creates bricks itself) or by journal scatters from the host path.
#pragma once
#include "host_maps.hpp"
#include "kernels_ivox_coop.hpp"
#include "kernels_ivox_update.hpp"

namespace fls {

struct IvoxImage {
    // ---- points ----
    DevBuf<float4> d_pts;
    size_t used = 0;  // slots in use on the device (includes per-voxel slack)
    size_t garbage = 0, n_pts_live = 0;
    // ---- hash-table form ----
    std::vector<HashEntry> table;
    DevBuf<HashEntry> d_table;
    unsigned mask = 0;
    bool want_hash = false;
    // ---- brick form ----
    bool have_bricks = false;
    FlatKeyMap brick_index;                      // packed brick key -> brick
    std::vector<unsigned long long> brick_keys;  // brick -> packed brick key
    std::vector<HashEntry> dir;                  // host copy of the device directory (same hash, same probing)
    unsigned dir_mask = 0;
    size_t n_bricks_cap = 0;
    bool dir_dirty = false;  // the host created bricks since the last upload
    DevBuf<HashEntry> d_dir;
    DevBuf<unsigned long long> d_brick_key;
    DevBuf<uint2> d_cells;
    DevBuf<unsigned> d_nbr;  // device-private neighbour-index cache of the update kernels (kernels_ivox_update.hpp IvoxUpdArrays::nbr)
    // ---- per-cell arrays of the device-side AddPoints: region capacity, LRU stamp, two scratch words ----
    DevBuf<unsigned char> d_cap_log2;
    DevBuf<unsigned long long> d_stamp;
    DevBuf<unsigned> d_pend, d_rank_mm;
    size_t meta_cells = 0;  // cells the four arrays are allocated (and initialised) for
    // ---- journal scatter ----
    struct PtUpd { unsigned slot; float x, y, z; int id; };
    struct CellUpd { unsigned long long idx; unsigned begin, count; };
    std::vector<PtUpd> pt_upd;
    std::vector<CellUpd> cell_upd;
    DevBuf<PtUpd> d_pt_upd;
    DevBuf<CellUpd> d_cell_upd;
    DevBuf<IvoxMetaRec> d_meta_rec;
    DevBuf<IvoxAliveRec> d_alive_rec;
    DevBuf<unsigned> d_counter;

    DevGrid dev() const { return DevGrid{d_table.p, d_pts.p, mask, unsigned(used)}; }
    BrickDir bricks() const {
        return have_bricks ? BrickDir{d_dir.p, dir_mask, d_cells.p, unsigned(n_bricks_cap)} : BrickDir{nullptr, 0u, nullptr, 0u};
    }
    size_t n_bricks() const { return brick_keys.size(); }

    static unsigned cap_for(size_t n) {
        unsigned c = 4;
        while (c < n) c <<= 1;
        return c;
    }
    static unsigned table_size_for(size_t n_keys) {
        size_t s = 1024;
        while (s < 2 * n_keys + 2) s <<= 1;
        return unsigned(s);
    }

    // ---- host side of the brick directory ----
    void dir_set(unsigned long long key, unsigned idx) {  // the entry of `key` (present or not) gets index idx
        unsigned h = brick_hash_of_key(key) & dir_mask;
        while (dir[h].key != kEmptyKey && dir[h].key != key) h = (h + 1) & dir_mask;
        dir[h] = HashEntry{key, idx, 0u};
    }
    static unsigned brick_hash_of_key(unsigned long long key) {
        int bx, by, bz;
        unpack_key(key, bx, by, bz);
        return brick_hash(bx, by, bz);
    }
    int find_brick(int bx, int by, int bz) const { return brick_index.find(pack_key(bx, by, bz)); }
    int get_brick(int bx, int by, int bz) {  // find or create; -1: the pool is full
        const unsigned long long key = pack_key(bx, by, bz);
        int v = brick_index.find(key);
        if (v >= 0) return v;
        if (brick_keys.size() >= n_bricks_cap) return -1;
        v = int(brick_keys.size());
        brick_keys.push_back(key);
        brick_index.insert(key, v);
        dir_set(key, unsigned(v));
        dir_dirty = true;
        return v;
    }
    // primary cell of voxel (x, y, z); false when its brick does not exist
    bool cell_index(int x, int y, int z, size_t& idx) const {
        const int b = find_brick(x >> kBrickLog, y >> kBrickLog, z >> kBrickLog);
        if (b < 0) return false;
        idx = size_t(b) * kBrickStride + brick_slab_index((x & (kBrickSide - 1)) + 1, (y & (kBrickSide - 1)) + 1, (z & (kBrickSide - 1)) + 1);
        return true;
    }
    // the bricks a voxel needs (its own + the neighbours whose halo mirrors it): created if missing; false when the pool is full
    bool ensure_bricks(int x, int y, int z) {
        const int bx = x >> kBrickLog, by = y >> kBrickLog, bz = z >> kBrickLog;
        bool ok = get_brick(bx, by, bz) >= 0;
        brick_for_each_mirror(x & (kBrickSide - 1), y & (kBrickSide - 1), z & (kBrickSide - 1),
                              [&](int dx, int dy, int dz) { if (ok) ok = get_brick(bx + dx, by + dy, bz + dz) >= 0; });
        return ok;
    }
    // records {cell, begin, count} of a voxel: primary + halo mirrors (all bricks must exist)
    void push_cell_records(int x, int y, int z, unsigned begin, unsigned count) {
        const int bx = x >> kBrickLog, by = y >> kBrickLog, bz = z >> kBrickLog;
        const int lx = x & (kBrickSide - 1), ly = y & (kBrickSide - 1), lz = z & (kBrickSide - 1);
        const int b = find_brick(bx, by, bz);
        cell_upd.push_back(CellUpd{(unsigned long long)b * kBrickStride + brick_slab_index(lx + 1, ly + 1, lz + 1), begin, count});
        brick_for_each_mirror(lx, ly, lz, [&](int dx, int dy, int dz) {
            const int nb = find_brick(bx + dx, by + dy, bz + dz);
            if (nb >= 0)
                cell_upd.push_back(CellUpd{(unsigned long long)nb * kBrickStride +
                                               brick_slab_index(lx + 1 - kBrickSide * dx, ly + 1 - kBrickSide * dy, lz + 1 - kBrickSide * dz), begin, count});
        });
    }
    void upload_directory(hipStream_t s) {
        d_dir.reserve(dir.size());
        d_brick_key.reserve(std::max<size_t>(n_bricks_cap, 1));
        FLS_HIP(hipMemcpyAsync(d_dir.p, dir.data(), dir.size() * sizeof(HashEntry), hipMemcpyHostToDevice, s));
        if (!brick_keys.empty())
            FLS_HIP(hipMemcpyAsync(d_brick_key.p, brick_keys.data(), brick_keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        FLS_HIP(hipStreamSynchronize(s));
        dir_dirty = false;
    }

    // what one brick of the pool costs: cells 8 B + the per-cell side arrays of the device AddPoints (cap 1 + stamp 8 + pend 4 + rank 4) per
    // slab cell, the reverse key and the neighbour-index row
    static constexpr size_t kPoolBrickBytes = size_t(kBrickStride) * (8 + 1 + 8 + 4 + 4) + 8 + 32 * 4;
    bool budget_exceeded = false;  // sticky: the owner switches to the hash-table form (P2PlaneIvoxMatcher::refresh_image)
    static size_t brick_budget_bytes() {
        static const size_t v = [] { const char* e = std::getenv("FLS_IVOX_BRICK_BUDGET_MB"); return size_t(e ? std::max(1L, std::atol(e)) : 16384L) << 20; }();
        return v;
    }
    // ---- full build from the host mirror ----
    void build_from_ivox(HostIvox& m, hipStream_t s, PinnedBuf<char>& stage) {
        // voxels in brick order (z, y, x), inside a brick in (z, y, x): spatially adjacent voxels get adjacent point buckets
        struct Ref { int x, y, z; HostIvox::Voxel* v; };
        std::vector<Ref> refs;
        refs.reserve(m.n_alive);
        size_t slots = 0;
        for (auto& v : m.pool) {
            if (!v.alive) continue;
            Ref r;
            unpack_key(v.key, r.x, r.y, r.z);
            r.v = &v;
            refs.push_back(r);
            slots += cap_for(v.pts.size());
        }
        auto order_key = [](const Ref& a) {
            return std::make_tuple(a.z >> kBrickLog, a.y >> kBrickLog, a.x >> kBrickLog, a.z, a.y, a.x);
        };
        std::sort(refs.begin(), refs.end(), [&](const Ref& a, const Ref& b) { return order_key(a) < order_key(b); });
        std::vector<Pt4> pts;
        pts.reserve(slots);
        if (!want_hash) {
            table.clear();
            mask = 0;
            // bricks: first the voxels' own (in sorted order), then the neighbours that only hold halo copies
            brick_index.clear();
            brick_keys.clear();
            // first pool: sized from the mirror (own bricks of the voxels, x3 for the halo-only neighbours and growth), not a fixed 8,192
            // bricks = 200 MB for an empty map
            size_t own = 0;
            for (size_t i = 0; i < refs.size(); ++i)
                if (i == 0 || (refs[i].x >> kBrickLog) != (refs[i - 1].x >> kBrickLog) || (refs[i].y >> kBrickLog) != (refs[i - 1].y >> kBrickLog) ||
                    (refs[i].z >> kBrickLog) != (refs[i - 1].z >> kBrickLog)) ++own;
            size_t cap_guess = std::max<size_t>(n_bricks_cap, 256);
            while (cap_guess < 3 * own) cap_guess *= 2;
            for (;;) {  // (size the pool before inserting: get_brick refuses beyond the capacity)
                if (cap_guess * kPoolBrickBytes > brick_budget_bytes()) {
                    // A spatially sparse map at the reference's 1e6-voxel capacity would need a brick per voxel (~25 KB each): beyond the
                    // budget the image takes the per-voxel hash table (host-maintained AddPoints) instead of failing an allocation
                    want_hash = true;
                    budget_exceeded = true;
                    n_bricks_cap = 0;
                    dir.clear(); brick_index.clear(); brick_keys.clear();
                    break;
                }
                n_bricks_cap = cap_guess;
                size_t ds = 4096;
                while (ds < 4 * n_bricks_cap) ds <<= 1;
                dir.assign(ds, HashEntry{kEmptyKey, kBrickPending, 0u});
                dir_mask = unsigned(ds - 1);
                brick_index.clear();
                brick_keys.clear();
                bool ok = true;
                for (const Ref& r : refs)
                    if (get_brick(r.x >> kBrickLog, r.y >> kBrickLog, r.z >> kBrickLog) < 0) { ok = false; break; }
                if (ok)
                    for (const Ref& r : refs)
                        if (!ensure_bricks(r.x, r.y, r.z)) { ok = false; break; }
                if (ok && 2 * brick_keys.size() <= n_bricks_cap) break;  // room for as many bricks again before the next rebuild
                cap_guess = std::max(cap_guess * 2, 2 * brick_keys.size());
            }
        }
        have_bricks = !want_hash;
        if (want_hash) {
            const unsigned ts = table_size_for(m.n_alive);
            mask = ts - 1;
            table.assign(ts, HashEntry{kEmptyKey, 0u, 0u});
        }
        cell_upd.clear();
        pt_upd.clear();
        for (const Ref& r : refs) {
            HostIvox::Voxel& v = *r.v;
            const unsigned beg = unsigned(pts.size()), cap = cap_for(v.pts.size());
            if (want_hash) {
                unsigned h = hash_key(v.key) & mask;
                while (table[h].key != kEmptyKey) h = (h + 1) & mask;
                table[h] = HashEntry{v.key, beg, unsigned(v.pts.size())};
            } else {
                push_cell_records(r.x, r.y, r.z, beg, unsigned(v.pts.size()));
            }
            pts.insert(pts.end(), v.pts.begin(), v.pts.end());
            pts.resize(size_t(beg) + cap, Pt4{0.f, 0.f, 0.f, -1});  // slack
            v.img_begin = beg; v.img_cap = cap; v.img_cnt = unsigned(v.pts.size());
        }
        used = pts.size();
        garbage = 0;
        n_pts_live = m.n_points;
        m.clear_journal();
        d_pts.reserve(used + used / 2 + (size_t(1) << 16));  // room for growth without reallocation
        if (used) FLS_HIP(hipMemcpyAsync(d_pts.p, pts.data(), used * sizeof(Pt4), hipMemcpyHostToDevice, s));
        if (want_hash) {
            d_table.reserve(table.size());
            FLS_HIP(hipMemcpyAsync(d_table.p, table.data(), table.size() * sizeof(HashEntry), hipMemcpyHostToDevice, s));
            FLS_HIP(hipStreamSynchronize(s));
            return;
        }
        d_cells.reserve(n_bricks_cap * kBrickStride);
        FLS_HIP(hipMemsetAsync(d_cells.p, 0, n_bricks_cap * kBrickStride * sizeof(uint2), s));
        d_nbr.reserve(n_bricks_cap * 32);
        FLS_HIP(hipMemsetAsync(d_nbr.p, 0xff, n_bricks_cap * 32 * sizeof(unsigned), s));  // (brick indices change with every build)
        meta_cells = 0;  // (the per-cell arrays follow in upload_update_meta)
        upload_directory(s);
        scatter_cell_records(s, stage);
        FLS_HIP(hipStreamSynchronize(s));
    }
    // ---- read-only replica (fls_replicas_*): the kNN side of another handle's image, device to device ----
    // points, directory and slabs (or the per-voxel table) and nothing of the AddPoints side: the handle that owns this copy serves
    // fls_match_batch only (P2PlaneIvoxMatcher::replicate_from).  `used_slots` / `n_bricks_live` are the SOURCE's current counts (its device
    // state while the device maintains the map); the source's stream must be idle.
    void clone_for_reading(const IvoxImage& src, size_t used_slots, size_t n_bricks_live, int src_device, int dst_device, hipStream_t s) {
        auto copy = [&](void* d, const void* q, size_t bytes) {
            if (!bytes) return;
            if (src_device == dst_device) FLS_HIP(hipMemcpyAsync(d, q, bytes, hipMemcpyDeviceToDevice, s));
            else FLS_HIP(hipMemcpyPeerAsync(d, dst_device, q, src_device, bytes, s));
        };
        want_hash = src.want_hash;
        have_bricks = src.have_bricks;
        used = used_slots; garbage = 0; n_pts_live = src.n_pts_live;
        table.clear(); brick_index.clear(); brick_keys.clear(); dir.clear(); dir_dirty = false; meta_cells = 0;
        cell_upd.clear(); pt_upd.clear();
        d_pts.reserve(std::max<size_t>(used_slots, 1));
        copy(d_pts.p, src.d_pts.p, used_slots * sizeof(float4));
        mask = src.mask;
        if (want_hash) {
            d_table.reserve(size_t(mask) + 1);
            copy(d_table.p, src.d_table.p, (size_t(mask) + 1) * sizeof(HashEntry));
        }
        dir_mask = src.dir_mask;
        n_bricks_cap = src.n_bricks_cap;
        if (have_bricks) {
            d_dir.reserve(size_t(dir_mask) + 1);
            copy(d_dir.p, src.d_dir.p, (size_t(dir_mask) + 1) * sizeof(HashEntry));
            d_cells.reserve(std::max<size_t>(n_bricks_live, 1) * kBrickStride);
            if (n_bricks_live == 0) FLS_HIP(hipMemsetAsync(d_cells.p, 0, kBrickStride * sizeof(uint2), s));
            copy(d_cells.p, src.d_cells.p, n_bricks_live * kBrickStride * sizeof(uint2));
        }
        FLS_HIP(hipStreamSynchronize(s));
    }

    // ---- the image as ONE flat buffer (round 5: fls_map_image_export / _import, include/fls_reg.h): header | points | [table] | [directory | cells].
    // What clone_for_reading copies between two handles of one process, laid out so that another PROCESS (another rank of a torch.distributed
    // job) can take it: the buffer is a device allocation the collective broadcasts in place (RCCL) or pinned host memory (gloo).  A receiver
    // becomes a read-only replica, exactly like a member of a replica set.
    struct FlatHeader {
        char magic[8];            // "FLSIMG01"
        unsigned kind, want_hash, have_bricks, is_first, use_dense, mask, dir_mask, pad;
        float resolution, pad_f;
        unsigned long long used, n_pts_live, n_bricks_live, n_bricks_cap, total_bytes;
    };
    static size_t flat_bytes(const FlatHeader& h) {
        size_t b = sizeof(FlatHeader) + size_t(h.used) * sizeof(float4);
        if (h.want_hash) b += (size_t(h.mask) + 1) * sizeof(HashEntry);
        if (h.have_bricks) b += (size_t(h.dir_mask) + 1) * sizeof(HashEntry) + size_t(h.n_bricks_live) * kBrickStride * sizeof(uint2);
        return b;
    }
    FlatHeader flat_header(size_t used_slots, size_t n_bricks_live) const {
        FlatHeader h{};
        std::memcpy(h.magic, "FLSIMG01", 8);
        h.want_hash = want_hash ? 1u : 0u; h.have_bricks = have_bricks ? 1u : 0u; h.use_dense = h.have_bricks;  // (the query takes the brick path exactly when the image carries bricks)
        h.mask = want_hash ? mask : 0u; h.dir_mask = have_bricks ? dir_mask : 0u;  // (a structure the image does not carry has no mask: flat_header_ok insists)
        h.used = used_slots; h.n_pts_live = n_pts_live; h.n_bricks_live = have_bricks ? n_bricks_live : 0; h.n_bricks_cap = n_bricks_cap;
        h.total_bytes = flat_bytes(h);
        return h;
    }
    // `dst` holds flat_bytes(h) bytes (device memory of THIS handle's device, or host memory); the owner's stream must be idle
    void export_flat(const FlatHeader& h, void* dst, bool dst_on_device, hipStream_t s) const {
        char* w = static_cast<char*>(dst);
        const hipMemcpyKind body = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        FLS_HIP(hipMemcpyAsync(w, &h, sizeof(h), dst_on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, s));
        size_t off = sizeof(h);
        auto put = [&](const void* q, size_t bytes) { if (bytes) FLS_HIP(hipMemcpyAsync(w + off, q, bytes, body, s)); off += bytes; };
        put(d_pts.p, size_t(h.used) * sizeof(float4));
        if (h.want_hash) put(d_table.p, (size_t(h.mask) + 1) * sizeof(HashEntry));
        if (h.have_bricks) { put(d_dir.p, (size_t(h.dir_mask) + 1) * sizeof(HashEntry)); put(d_cells.p, size_t(h.n_bricks_live) * kBrickStride * sizeof(uint2)); }
        FLS_HIP(hipStreamSynchronize(s));
    }
    // the header is untrusted (it comes off a broadcast): every size is bounded by the payload before anything is allocated
    static bool flat_header_ok(const FlatHeader& h, size_t n_bytes) {
        if (std::memcmp(h.magic, "FLSIMG01", 8) != 0) return false;
        auto pow2m1 = [](unsigned m) { return (m & (m + 1u)) == 0u; };
        if (h.want_hash > 1u || h.have_bricks > 1u || (!h.want_hash && !h.have_bricks)) return false;
        if (!pow2m1(h.mask) || !pow2m1(h.dir_mask)) return false;
        // a table the image does not carry has no mask either (`mask != 0` without the table made the query take the hash path into a null table),
        // and the dense (brick) path is exactly "the image has bricks"
        if ((!h.want_hash && h.mask != 0u) || (!h.have_bricks && (h.dir_mask != 0u || h.n_bricks_live != 0ull)) || h.use_dense != h.have_bricks) return false;
        const unsigned long long cap = (unsigned long long)n_bytes / 8ull;  // no array can hold more records than the payload has 8-byte words
        if (h.used > cap || h.n_bricks_live > cap / kBrickStride + 1 || h.n_bricks_cap > (1ull << 31) || h.n_bricks_live > h.n_bricks_cap || h.n_pts_live > h.used) return false;
        if ((unsigned long long)h.mask > cap || (unsigned long long)h.dir_mask > cap) return false;
        return h.total_bytes == (unsigned long long)n_bytes && flat_bytes(h) == n_bytes;
    }
    void import_flat(const FlatHeader& h, const void* src, bool src_on_device, hipStream_t s) {
        const char* r = static_cast<const char*>(src);
        const hipMemcpyKind body = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        want_hash = h.want_hash != 0; have_bricks = h.have_bricks != 0;
        used = size_t(h.used); garbage = 0; n_pts_live = size_t(h.n_pts_live);
        table.clear(); brick_index.clear(); brick_keys.clear(); dir.clear(); dir_dirty = false; meta_cells = 0;
        cell_upd.clear(); pt_upd.clear();
        // (the brick bound the query kernels receive is what THIS handle allocates -- the live bricks -- not the exporter's pool size: ADVICE r5)
        mask = h.mask; dir_mask = h.dir_mask; n_bricks_cap = have_bricks ? size_t(h.n_bricks_live) : 0;
        size_t off = sizeof(h);
        auto get = [&](void* d, size_t bytes) { if (bytes) FLS_HIP(hipMemcpyAsync(d, r + off, bytes, body, s)); off += bytes; };
        d_pts.reserve(std::max<size_t>(used, 1));
        get(d_pts.p, used * sizeof(float4));
        if (want_hash) { d_table.reserve(size_t(mask) + 1); get(d_table.p, (size_t(mask) + 1) * sizeof(HashEntry)); }
        if (have_bricks) {
            d_dir.reserve(size_t(dir_mask) + 1);
            get(d_dir.p, (size_t(dir_mask) + 1) * sizeof(HashEntry));
            const size_t nb = size_t(h.n_bricks_live);
            d_cells.reserve(std::max<size_t>(nb, 1) * kBrickStride);
            if (nb == 0) FLS_HIP(hipMemsetAsync(d_cells.p, 0, kBrickStride * sizeof(uint2), s));
            get(d_cells.p, nb * kBrickStride * sizeof(uint2));
        }
        FLS_HIP(hipStreamSynchronize(s));
    }

    // cell_upd (and pt_upd) -> the device image
    void scatter_cell_records(hipStream_t s, PinnedBuf<char>& stage) {
        const size_t np = pt_upd.size(), nc = cell_upd.size();
        if (np + nc == 0) return;
        d_pt_upd.reserve(std::max<size_t>(np, 1));
        d_cell_upd.reserve(std::max<size_t>(nc, 1));
        stage.reserve(np * sizeof(PtUpd) + nc * sizeof(CellUpd) + 16);
        std::memcpy(stage.p, pt_upd.data(), np * sizeof(PtUpd));
        char* cpos = stage.p + np * sizeof(PtUpd);
        std::memcpy(cpos, cell_upd.data(), nc * sizeof(CellUpd));
        if (np) FLS_HIP(hipMemcpyAsync(d_pt_upd.p, stage.p, np * sizeof(PtUpd), hipMemcpyHostToDevice, s));
        if (nc) FLS_HIP(hipMemcpyAsync(d_cell_upd.p, cpos, nc * sizeof(CellUpd), hipMemcpyHostToDevice, s));
        const size_t mx = std::max(np, nc);
        hipLaunchKernelGGL(ivox_apply_updates_kernel, dim3(unsigned((mx + 255) / 256)), dim3(256), 0, s, (const PtUpdDev*)d_pt_upd.p, int(np),
                           (const CellUpdDev*)d_cell_upd.p, int(nc), d_pts.p, d_cells.p);
        FLS_HIP(hipGetLastError());
        FLS_HIP(hipStreamSynchronize(s));  // the staging buffer is reused by the next update
    }

    // Per-cell arrays of the device-side AddPoints, regenerated from the mirror (which must be in sync with the image: right after
    // build_from_ivox or a journal update).  LRU stamps 1..n_alive from the tail (oldest) to the head.  Returns the stamp base.
    unsigned long long upload_update_meta(const HostIvox& m, hipStream_t s, PinnedBuf<char>& stage) {
        const size_t ncell = n_bricks_cap * kBrickStride;
        d_cap_log2.reserve(ncell);
        d_stamp.reserve(ncell);
        d_pend.reserve(ncell);
        d_rank_mm.reserve(ncell);
        meta_cells = ncell;
        FLS_HIP(hipMemsetAsync(d_cap_log2.p, 0, ncell, s));
        FLS_HIP(hipMemsetAsync(d_stamp.p, 0, ncell * sizeof(unsigned long long), s));
        FLS_HIP(hipMemsetAsync(d_pend.p, 0, ncell * sizeof(unsigned), s));
        FLS_HIP(hipMemsetAsync(d_rank_mm.p, 0xff, ncell * sizeof(unsigned), s));
        stage.reserve(m.n_alive * sizeof(IvoxMetaRec) + 16);
        IvoxMetaRec* rec = reinterpret_cast<IvoxMetaRec*>(stage.p);
        unsigned long long t = 0;
        size_t n = 0;
        for (int v = m.tail; v >= 0; v = m.pool[v].prev) {
            int x, y, z;
            unpack_key(m.pool[v].key, x, y, z);
            size_t idx;
            if (!cell_index(x, y, z, idx)) continue;
            unsigned l = 0;
            while ((1u << l) < m.pool[v].img_cap) ++l;
            rec[n++] = IvoxMetaRec{unsigned(idx), l, ++t};
        }
        if (n) {
            d_meta_rec.reserve(n);
            FLS_HIP(hipMemcpyAsync(d_meta_rec.p, rec, n * sizeof(IvoxMetaRec), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(ivox_meta_scatter_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, (const IvoxMetaRec*)d_meta_rec.p, unsigned(n), d_cap_log2.p,
                               d_stamp.p);
            FLS_HIP(hipGetLastError());
        }
        FLS_HIP(hipStreamSynchronize(s));
        return t;
    }

    // Collect the journal of `m` into update records.  Returns false when a full rebuild is required.
    bool collect_incremental(HostIvox& m) {
        if (!have_bricks || want_hash) return false;
        pt_upd.clear();
        cell_upd.clear();
        for (unsigned long long key : m.evicted_keys) {
            int x, y, z;
            unpack_key(key, x, y, z);
            size_t idx;
            if (cell_index(x, y, z, idx)) push_cell_records(x, y, z, 0u, 0u);
        }
        size_t new_used = used, new_garbage = garbage;
        for (int vi : m.touched) {
            HostIvox::Voxel& v = m.pool[vi];
            if (!v.alive) continue;  // created and evicted inside the same batch
            int x, y, z;
            unpack_key(v.key, x, y, z);
            if (!ensure_bricks(x, y, z)) return false;  // brick pool full: rebuild with a larger one
            const unsigned cnt = unsigned(v.pts.size());
            unsigned from = v.img_cnt;
            if (cnt > v.img_cap) {  // relocate the bucket to the end of the array, twice the room
                const unsigned cap = cap_for(cnt);
                if (new_used + cap > d_pts.cap) return false;
                new_garbage += v.img_cap;
                v.img_begin = unsigned(new_used);
                v.img_cap = cap;
                new_used += cap;
                from = 0;
            }
            for (unsigned k = from; k < cnt; ++k) pt_upd.push_back(PtUpd{v.img_begin + k, v.pts[k].x, v.pts[k].y, v.pts[k].z, v.pts[k].id});
            v.img_cnt = cnt;
            push_cell_records(x, y, z, v.img_begin, cnt);
        }
        // a voxel evicted and re-created inside one batch yields two records for the same cell: the scatter kernel
        // writes records in parallel, so keep only the LAST record per cell (stable sort, then unique from the back)
        std::stable_sort(cell_upd.begin(), cell_upd.end(), [](const CellUpd& a, const CellUpd& b) { return a.idx < b.idx; });
        size_t w = 0;
        for (size_t r = 0; r < cell_upd.size(); ++r) {
            if (r + 1 < cell_upd.size() && cell_upd[r + 1].idx == cell_upd[r].idx) continue;
            cell_upd[w++] = cell_upd[r];
        }
        cell_upd.resize(w);
        if (new_garbage * 2 > new_used && new_used > (size_t(1) << 20)) return false;  // compact
        used = new_used;
        garbage = new_garbage;
        n_pts_live = m.n_points;
        m.clear_journal();
        return true;
    }

    // The device's bricks (it creates them itself in device mode) back into the host copy of the directory.
    void download_directory(size_t n_bricks_dev, hipStream_t s) {
        n_bricks_dev = std::min(n_bricks_dev, n_bricks_cap);
        brick_keys.resize(n_bricks_dev);
        FLS_HIP(hipMemcpyAsync(dir.data(), d_dir.p, dir.size() * sizeof(HashEntry), hipMemcpyDeviceToHost, s));
        if (n_bricks_dev) FLS_HIP(hipMemcpyAsync(brick_keys.data(), d_brick_key.p, n_bricks_dev * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        brick_index.clear();
        for (size_t b = 0; b < n_bricks_dev; ++b) brick_index.insert(brick_keys[b], int(b));
        // (entries a refused batch left with kBrickInvalid / kBrickPending: re-pointed by dir_set when the host creates that brick)
        dir_dirty = false;
    }
};

}  // namespace fls
