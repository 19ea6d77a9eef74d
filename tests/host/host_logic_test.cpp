// Host-logic checks of the PRODUCT's C++ bookkeeping that need no GPU (compiled with hipcc, run on the CPU):
//   * HostIvox insert / LRU-evict semantics vs a straightforward list+map model and vs the oracle's iVox
//   * voxel_grid (PCL VoxelGrid semantics) vs the oracle's
//   * IvoxImage::collect_incremental invariants (disjoint slot regions, unique cell records, eviction records, halo mirrors of the brick image)
// No HIP runtime call is made: only host members are touched.
#include "../../funny_lidar_slam_amd/csrc/host_maps.hpp"
#include "../../funny_lidar_slam_amd/csrc/ivox_image.hpp"
#include "../../funny_lidar_slam_amd/csrc/host_math.hpp"
#include "../../funny_lidar_slam_amd/csrc/replicas.hpp"
#include "../../oracle/flo_api.h"
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <random>
#include <set>
#include <thread>

using namespace fls;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::vector<PtI> random_cloud(std::mt19937& rng, size_t n, float extent) {
    std::uniform_real_distribution<float> u(-extent, extent), w(-2.f, 2.f);
    std::vector<PtI> c(n);
    for (auto& p : c) { p.x = u(rng); p.y = u(rng); p.z = w(rng); p.i = u(rng); }
    return c;
}

int main() {
    std::mt19937 rng(12345);
    // ---- 1. HostIvox vs a model with the reference's rule (ivox_map.cpp:122-143) -----------------------
    {
        HostIvox iv;
        iv.capacity = 300;
        std::list<std::pair<unsigned long long, std::vector<int>>> cache;  // front = most recent
        std::map<unsigned long long, decltype(cache)::iterator> index;
        int next_id = 0;
        for (int round = 0; round < 6; ++round) {
            const auto cloud = random_cloud(rng, 2000, 6.0f + round);
            CHECK(iv.add_points(cloud.data(), cloud.size()) == FLS_OK);
            for (const PtI& p : cloud) {
                int kx, ky, kz;
                HostIvox::key_of(p.x, p.y, p.z, iv.inv_resolution, kx, ky, kz);
                const unsigned long long key = pack_key(kx, ky, kz);
                auto it = index.find(key);
                if (it == index.end()) {
                    cache.push_front({key, {next_id++}});
                    index[key] = cache.begin();
                    if (index.size() >= iv.capacity) { index.erase(cache.back().first); cache.pop_back(); }
                } else {
                    it->second->second.push_back(next_id++);
                    cache.splice(cache.begin(), cache, it->second);
                    index[key] = cache.begin();
                }
            }
            CHECK(iv.n_alive == index.size());
            size_t pts = 0;
            for (auto& kv : cache) pts += kv.second.size();
            CHECK(iv.n_points == pts);
            // LRU order and contents
            int v = iv.head;
            for (auto& kv : cache) {
                CHECK(v >= 0 && iv.pool[v].alive && iv.pool[v].key == kv.first && iv.pool[v].pts.size() == kv.second.size());
                for (size_t k = 0; k < kv.second.size(); ++k) CHECK(iv.pool[v].pts[k].id == kv.second[k]);
                v = iv.pool[v].next;
            }
            CHECK(v == -1);
        }
        int x, y, z;
        unpack_key(pack_key(-5, 1048000, -1048000), x, y, z);
        CHECK(x == -5 && y == 1048000 && z == -1048000);
        PtI far{6.0e5f, 0.f, 0.f, 0.f};  // 1.2e6 voxels out: beyond the 21-bit key range
        CHECK(iv.add_points(&far, 1) == FLS_ERR_RANGE);
    }
    // ---- 2. HostIvox vs the oracle's iVox through its C API (sizes after inserts) -----------------------
    {
        flo_params p{};
        p.struct_size = sizeof(p);
        p.max_iterations = 1; p.point_to_planar_thres = 0.1; p.position_converge_thres = 0.005; p.rotation_converge_thres = 0.001;
        void* o = flo_create(FLO_P2PLANE_IVOX, &p);
        CHECK(o != nullptr);
        flo_set_ivox_capacity(o, 500);
        HostIvox iv;
        iv.capacity = 500;
        const auto cloud = random_cloud(rng, 20000, 9.0f);
        std::vector<float> flat(cloud.size() * 4);
        for (size_t i = 0; i < cloud.size(); ++i) { flat[4 * i] = cloud[i].x; flat[4 * i + 1] = cloud[i].y; flat[4 * i + 2] = cloud[i].z; flat[4 * i + 3] = cloud[i].i; }
        flo_add_cloud(o, flat.data(), cloud.size(), nullptr, 0, 4);
        CHECK(iv.add_points(cloud.data(), cloud.size()) == FLS_OK);
        CHECK(iv.n_points == flo_map_size(o, 0));
        CHECK(iv.n_alive == flo_map_voxels(o));
        flo_destroy(o);
    }
    // ---- 3. voxel_grid vs the oracle -------------------------------------------------------------------
    for (float leaf : {0.2f, 0.4f, 1.0f}) {
        const auto cloud = random_cloud(rng, 30000, 15.0f);
        std::vector<float> flat(cloud.size() * 4), out(cloud.size() * 4);
        for (size_t i = 0; i < cloud.size(); ++i) { flat[4 * i] = cloud[i].x; flat[4 * i + 1] = cloud[i].y; flat[4 * i + 2] = cloud[i].z; flat[4 * i + 3] = cloud[i].i; }
        const size_t m = flo_voxel_grid(flat.data(), cloud.size(), 4, leaf, out.data());
        const auto vg = voxel_grid(cloud, leaf);
        CHECK(vg.size() == m);
        for (size_t i = 0; i < m; ++i)
            CHECK(vg[i].x == out[4 * i] && vg[i].y == out[4 * i + 1] && vg[i].z == out[4 * i + 2] && vg[i].i == out[4 * i + 3]);
    }
    CHECK(voxel_grid(std::vector<PtI>(), 0.4f).empty());
    // ---- 3b. the exact restatement of std::sort (host_parallel.hpp): the SAME permutation as libstdc++'s introsort on
    //          records that compare by their key only (heavy ties), sequential form and pool form, all sizes --------------
    {
        struct Rec { unsigned idx, pt; bool operator<(const Rec& o) const { return idx < o.idx; } };
        HostPool& pool = HostPool::get();
        for (int round = 0; round < 260; ++round) {
            const size_t n = round < 40 ? size_t(round) : (round < 200 ? size_t(rng() % 5000) : size_t(20000 + rng() % 200000));
            const unsigned keys = round % 5 == 0 ? 3u : (round % 5 == 1 ? unsigned(n / 2 + 1) : (round % 5 == 2 ? 0xffffffffu : unsigned(n / 7 + 1)));
            std::vector<Rec> a(n);
            for (size_t i = 0; i < n; ++i) a[i] = Rec{keys == 0xffffffffu ? unsigned(rng()) : unsigned(rng() % keys), unsigned(i)};
            if (round % 7 == 3) std::sort(a.begin(), a.end(), [](const Rec& x, const Rec& y) { return x.idx < y.idx || (x.idx == y.idx && x.pt < y.pt); });  // already sorted
            if (round % 7 == 4) std::reverse(a.begin(), a.end());
            std::vector<Rec> ref = a, s1 = a, s2 = a;
            std::sort(ref.begin(), ref.end());
            CHECK(exact_sort_sequential(s1.data(), s1.data() + n));
            for (size_t i = 0; i < n; ++i) CHECK(s1[i].idx == ref[i].idx && s1[i].pt == ref[i].pt);
            ExactSortShared<Rec> sh;
            sh.reset(s2.data(), s2.data() + n);
            const bool ran = pool.run([&](HostPool::Region& reg) { reg.phase(size_t(pool.threads()), [&](size_t) { exact_sort_worker(sh); }); });
            if (ran) {
                CHECK(!sh.failed.load());
                for (size_t i = 0; i < n; ++i) CHECK(s2[i].idx == ref[i].idx && s2[i].pt == ref[i].pt);
            } else {
                CHECK(pool.threads() <= 1);
            }
        }
    }
    // ---- 3c. the pooled VoxelGrid == the sequential one, bit for bit (sizes that take the pool path; NaN points; the
    //          "leaf size too small" copy) -------------------------------------------------------------------------------
    for (int round = 0; round < 6; ++round) {
        auto cloud = random_cloud(rng, 60000 + 30000 * size_t(round), round % 2 ? 60.0f : 8.0f);
        if (round == 2) for (size_t i = 0; i < cloud.size(); i += 97) cloud[i].y = std::nanf("");
        const float leaf = round == 5 ? 0.0005f : (round % 3 == 0 ? 0.2f : 0.5f);
        const auto a = voxel_grid(cloud, leaf), b = voxel_grid_sequential(cloud, leaf);
        CHECK(a.size() == b.size());
        CHECK(a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(PtI)) == 0 || round == 2);
        if (round == 2) for (size_t i = 0; i < a.size(); ++i) CHECK(std::memcmp(&a[i], &b[i], sizeof(PtI)) == 0 || (a[i].x != a[i].x && b[i].x != b[i].x));
        // the caller's strided cloud read in place (packed xyzi rows and 32-byte PCL points) == the copied cloud
        for (int stride : {4, 8}) {
            std::vector<float> raw(cloud.size() * size_t(stride), 0.f);
            for (size_t i = 0; i < cloud.size(); ++i) {
                float* q = &raw[i * size_t(stride)];
                q[0] = cloud[i].x; q[1] = cloud[i].y; q[2] = cloud[i].z; q[stride >= 8 ? 4 : 3] = cloud[i].i;
            }
            const auto c = voxel_grid_strided(raw.data(), cloud.size(), stride, leaf);
            CHECK(c.size() == b.size());
            if (round != 2) CHECK(c.empty() || std::memcmp(c.data(), b.data(), c.size() * sizeof(PtI)) == 0);
        }
    }
    // ---- 3d. several threads filtering at once (batch lanes): one of them gets the pool, the others run the sequential code;
    //          every result is the sequential one -----------------------------------------------------------------------
    {
        std::vector<std::vector<PtI>> clouds, want(6), got(6);
        for (int t = 0; t < 6; ++t) clouds.push_back(random_cloud(rng, 70000 + 9000 * size_t(t), 12.0f));
        for (int t = 0; t < 6; ++t) want[size_t(t)] = voxel_grid_sequential(clouds[size_t(t)], 0.3f);
        for (int rep = 0; rep < 4; ++rep) {
            std::vector<std::thread> th;
            for (int t = 0; t < 6; ++t) th.emplace_back([&, t] { got[size_t(t)] = voxel_grid(clouds[size_t(t)], 0.3f); });
            for (auto& x : th) x.join();
            for (int t = 0; t < 6; ++t) {
                CHECK(got[size_t(t)].size() == want[size_t(t)].size());
                CHECK(std::memcmp(got[size_t(t)].data(), want[size_t(t)].data(), want[size_t(t)].size() * sizeof(PtI)) == 0);
            }
        }
    }
    // ---- 3e. the Gauss-Newton fast path (host_math.hpp::ldlt_solve6, the same code the device tail runs): agrees with the oracle's
    //          restatement of Eigen's FullPivHouseholderQR on well-conditioned normal equations, declines singular / indefinite / NaN ones
    {
        std::normal_distribution<double> nd(0.0, 1.0);
        int solved = 0;
        for (int round = 0; round < 400; ++round) {
            const int rows = 6 + int(rng() % 300);
            double H[36] = {0}, g[6] = {0};
            for (int r = 0; r < rows; ++r) {
                double J[6];
                for (int a = 0; a < 6; ++a) J[a] = nd(rng) * (a < 3 ? 30.0 : 1.0);  // rotation columns ~ range, translation columns ~ 1
                if (round % 10 == 9) J[5] = 0.0;                                       // an unobserved direction: singular
                const double res = nd(rng) * 0.05;
                for (int a = 0; a < 6; ++a) { g[a] += -J[a] * res; for (int b = 0; b < 6; ++b) H[a + 6 * b] += J[a] * J[b]; }
            }
            double x[6], xr[6];
            const bool ok = hm::ldlt_solve6(H, g, x);
            if (round % 10 == 9) { CHECK(!ok); continue; }
            CHECK(ok);
            flo_fullpiv_qr_solve_6(H, g, xr);
            double num = 0.0, den = 0.0;
            for (int a = 0; a < 6; ++a) { num += (x[a] - xr[a]) * (x[a] - xr[a]); den += xr[a] * xr[a]; }
            CHECK(std::sqrt(num) <= 1e-10 * std::sqrt(den) + 1e-18);
            ++solved;
        }
        CHECK(solved == 360);
        double Hn[36] = {0}, gn[6] = {1, 1, 1, 1, 1, 1}, xn[6];
        for (int a = 0; a < 6; ++a) Hn[a + 6 * a] = 1.0;
        Hn[0] = std::nan("");
        CHECK(!hm::ldlt_solve6(Hn, gn, xn));
        Hn[0] = -1.0;
        CHECK(!hm::ldlt_solve6(Hn, gn, xn));
        Hn[0] = 1.0; gn[2] = std::nan("");
        CHECK(!hm::ldlt_solve6(Hn, gn, xn));
    }
    // ---- 4. brick-image bookkeeping on the host (ivox_image.hpp): journal records, halo mirrors, brick creation ---------------
    {
        HostIvox iv;
        iv.capacity = 400;
        IvoxImage img;
        img.want_hash = false;
        img.have_bricks = true;
        img.n_bricks_cap = 64;  // small pool: the test also reaches "pool full -> rebuild"
        img.dir.assign(256, HashEntry{kEmptyKey, kBrickPending, 0u});
        img.dir_mask = 255;
        img.d_pts.cap = size_t(1) << 22;  // pretend the device array is large (never dereferenced here)
        std::map<unsigned long long, std::pair<unsigned, unsigned>> cells;  // the image's cell array as a map {cell -> begin, count}
        bool pool_filled = false;
        for (int round = 0; round < 8 && !pool_filled; ++round) {
            const auto cloud = random_cloud(rng, 1500, 4.0f + 2.0f * round);
            CHECK(iv.add_points(cloud.data(), cloud.size()) == FLS_OK);
            if (!img.collect_incremental(iv)) { pool_filled = true; CHECK(img.n_bricks() == img.n_bricks_cap); break; }
            CHECK(iv.touched.empty() && iv.evicted_keys.empty());
            std::set<unsigned long long> seen_cells;
            for (const auto& c : img.cell_upd) CHECK(seen_cells.insert(c.idx).second);  // unique per cell
            std::set<unsigned> seen_slots;
            for (const auto& u : img.pt_upd) CHECK(seen_slots.insert(u.slot).second);    // unique per slot
            for (const auto& c : img.cell_upd) { if (c.count) cells[c.idx] = {c.begin, c.count}; else cells.erase(c.idx); }
            // regions of alive voxels are disjoint and hold exactly the voxel's points
            std::vector<std::pair<unsigned, unsigned>> regions;
            size_t alive = 0, expected_cells = 0;
            for (const auto& v : iv.pool) {
                if (!v.alive) continue;
                ++alive;
                CHECK(v.img_cnt == v.pts.size() && v.img_cnt <= v.img_cap && v.img_begin + v.img_cap <= img.used);
                regions.push_back({v.img_begin, v.img_begin + v.img_cap});
                // the brick invariant: the voxel's primary cell AND every halo copy hold {begin, count}; all those bricks exist
                int x, y, z;
                unpack_key(v.key, x, y, z);
                size_t idx;
                CHECK(img.cell_index(x, y, z, idx));
                CHECK(cells.count(idx) && cells[idx] == std::make_pair(v.img_begin, v.img_cnt));
                ++expected_cells;
                const int bx = x >> kBrickLog, by = y >> kBrickLog, bz = z >> kBrickLog, lx = x & 7, ly = y & 7, lz = z & 7;
                bool ok = true;
                brick_for_each_mirror(lx, ly, lz, [&](int dx, int dy, int dz) {
                    const int nb = img.find_brick(bx + dx, by + dy, bz + dz);
                    if (nb < 0) { ok = false; return; }
                    const unsigned long long m = (unsigned long long)nb * kBrickStride + brick_slab_index(lx + 1 - 8 * dx, ly + 1 - 8 * dy, lz + 1 - 8 * dz);
                    if (!cells.count(m) || cells[m] != std::make_pair(v.img_begin, v.img_cnt)) ok = false;
                    ++expected_cells;
                });
                CHECK(ok);
            }
            std::sort(regions.begin(), regions.end());
            for (size_t i = 1; i < regions.size(); ++i) CHECK(regions[i - 1].second <= regions[i].first);
            CHECK(cells.size() == expected_cells);  // nothing but the alive voxels' cells and mirrors is set: every evicted voxel was cleared everywhere
            // the host directory answers like the device's: every brick reachable by linear probing from its hash
            for (size_t b2 = 0; b2 < img.n_bricks(); ++b2) {
                unsigned h = IvoxImage::brick_hash_of_key(img.brick_keys[b2]) & img.dir_mask;
                while (img.dir[h].key != img.brick_keys[b2]) { CHECK(img.dir[h].key != kEmptyKey); h = (h + 1) & img.dir_mask; }
                CHECK(img.dir[h].begin == b2);
            }
        }
        CHECK(pool_filled);  // 64 bricks cannot hold the later rounds: collect_incremental asked for a rebuild
        CHECK(IvoxImage::cap_for(0) == 4 && IvoxImage::cap_for(5) == 8 && IvoxImage::cap_for(8) == 8 && IvoxImage::cap_for(9) == 16);
        // slab geometry: interior cells are exactly the 8^3 block at stored coordinates 1..8; mirrors of a corner voxel are its 3 faces + 3 edges
        int n_int = 0, sx, sy, sz;
        for (unsigned l = 0; l < kBrickStride; ++l) n_int += brick_slab_interior(l, sx, sy, sz) ? 1 : 0;
        CHECK(n_int == 512);
        int n_m = 0;
        bool shape_ok = true;
        brick_for_each_mirror(0, 7, 0, [&](int dx, int dy, int dz) { ++n_m; shape_ok = shape_ok && dx <= 0 && dy >= 0 && dz <= 0 && (dx || dy || dz) && !(dx && dy && dz); });
        CHECK(n_m == 6 && shape_ok);
        n_m = 0; brick_for_each_mirror(3, 7, 4, [&](int dx, int dy, int dz) { ++n_m; shape_ok = shape_ok && dx == 0 && dy == 1 && dz == 0; }); CHECK(n_m == 1 && shape_ok);
        n_m = 0; brick_for_each_mirror(3, 5, 4, [&](int, int, int) { ++n_m; }); CHECK(n_m == 0);
    }
    // ---- the job partition of the native replica set (csrc/replicas.hpp) = funny_lidar_slam_amd/batch.py::partition: contiguous blocks,
    // sizes differ by at most one, the first n % world ranks take the extra job, every job exactly once
    for (size_t world = 1; world <= 9; ++world)
        for (size_t n = 0; n <= 40; ++n) {
            size_t next = 0;
            for (size_t r = 0; r < world; ++r) {
                size_t b = 0, e = 0;
                fls_replicas::block(n, world, r, b, e);
                CHECK(b == next && e >= b);
                const size_t base = n / world, extra = n % world;
                CHECK(e - b == base + (r < extra ? 1 : 0));
                next = e;
            }
            CHECK(next == n);
        }
    // ---- flat device image header (fls_map_image_import): the header comes off a broadcast, every size must be bounded by the payload ----
    {
        IvoxImage img;
        img.have_bricks = true; img.want_hash = false; img.mask = 0; img.dir_mask = 4095; img.n_bricks_cap = 64; img.n_pts_live = 900;
        IvoxImage::FlatHeader h = img.flat_header(1000, 10);
        const size_t n = size_t(h.total_bytes);
        CHECK(n == sizeof(IvoxImage::FlatHeader) + 1000 * 16 + 4096 * 16 + 10 * size_t(kBrickStride) * 8);
        CHECK(IvoxImage::flat_header_ok(h, n));
        CHECK(!IvoxImage::flat_header_ok(h, n - 8));                                   // truncated payload
        { auto g = h; g.magic[0] = 'X'; CHECK(!IvoxImage::flat_header_ok(g, n)); }
        { auto g = h; g.used = ~0ull / 4; CHECK(!IvoxImage::flat_header_ok(g, n)); }      // would wrap a size computation
        { auto g = h; g.n_bricks_live = 65; CHECK(!IvoxImage::flat_header_ok(g, n)); }   // more live bricks than the pool
        { auto g = h; g.dir_mask = 4094; CHECK(!IvoxImage::flat_header_ok(g, n)); }      // not 2^k - 1
        { auto g = h; g.have_bricks = 0; CHECK(!IvoxImage::flat_header_ok(g, n)); }      // neither form
        { auto g = h; g.n_pts_live = 1001; CHECK(!IvoxImage::flat_header_ok(g, n)); }
        { auto g = h; g.total_bytes += 16; CHECK(!IvoxImage::flat_header_ok(g, n)); }
        //  a mask without its table, a dense flag without bricks: the query would probe a table / slab the import never allocated
        { auto g = h; g.mask = 1023; CHECK(!IvoxImage::flat_header_ok(g, n)); }
        { auto g = h; g.use_dense = 0; CHECK(!IvoxImage::flat_header_ok(g, n)); }
        {
            IvoxImage t; t.have_bricks = false; t.want_hash = true; t.mask = 1023; t.dir_mask = 4095 /* stale: an image that fell back to the table */; t.n_pts_live = 10;
            IvoxImage::FlatHeader g = t.flat_header(16, 0);
            CHECK(g.dir_mask == 0u && g.use_dense == 0u && IvoxImage::flat_header_ok(g, size_t(g.total_bytes)));
            g.dir_mask = 4095; CHECK(!IvoxImage::flat_header_ok(g, size_t(g.total_bytes)));
        }
        CHECK(IvoxImage::kPoolBrickBytes > 25000 && IvoxImage::kPoolBrickBytes < 27000);
    }
    std::printf("host logic ok\n");
    return 0;
}
