// matcher_ndt.hpp -- host side of FLS_INCREMENTAL_NDT, the replacement of IncrementalNDT
// (include/registration/incremental_ndt.h:16-383).
//   AddCloudToLocalMap :182-227  voxel insert with LRU eviction + UpdateVoxel (:130-179) on the host;
//                                the estimated voxels are flattened into a device hash table
//                                {key -> slot} with mu[slot], info[slot] in FP64.
//   Match              :229-337  device-resident loop: ndt_kernel (7-voxel Mahalanobis scoring,
//                                FP64) + gn_solve_lu_kernel (mode 1).
#pragma once
#include "matcher_base.hpp"
#include "host_math.hpp"
#include "device_voxelgrid.hpp"
#include "kernels_ndt_update.hpp"
#include "kernels_knn.hpp"
#include "fitness_host.hpp"
#include <chrono>
#include <list>

namespace fls {

struct NdtMatcher final : fls_matcher {
    struct Voxel {
        int kx, ky, kz;
        std::vector<double> pts;  // xyz triples waiting for the next estimate
        double mu[3] = {0, 0, 0}, sigma[9] = {0}, info[9] = {0};
        bool estimated = false;
        int num_points = 0;
        int vid = 0;
        int prev = -1, next = -1;  // LRU list (head = most recently touched)
        int slot = -1;             // row of the device image (estimated voxels only)
        unsigned epoch = 0;        // last add_cloud call that touched it
        bool dirty = false;        // mu / info changed in this call
    };
    // std::list + unordered_map of the reference (incremental_ndt.h:352-353) as a pool with an intrusive LRU list and an
    // open-addressing key map: same sequence of creations, touches and evictions, no node allocation per point
    std::vector<Voxel> pool;
    std::vector<int> free_ix;
    int lru_head = -1, lru_tail = -1;
    size_t n_alive = 0;
    FlatKeyMap grids;  // key -> pool index
    unsigned epoch = 0;
    void lru_unlink(int v) {
        Voxel& x = pool[v];
        if (x.prev >= 0) pool[x.prev].next = x.next; else lru_head = x.next;
        if (x.next >= 0) pool[x.next].prev = x.prev; else lru_tail = x.prev;
        x.prev = x.next = -1;
    }
    void lru_push_front(int v) {
        Voxel& x = pool[v];
        x.prev = -1;
        x.next = lru_head;
        if (lru_head >= 0) pool[lru_head].prev = v; else lru_tail = v;
        lru_head = v;
    }
    bool flag_first_scan = true;
    int next_vid = 0;
    double inv_voxel = 1.0;

    // device image
    std::vector<HashEntry> h_table;
    std::vector<double> h_mu, h_info;
    std::vector<int> h_vid;
    DevBuf<HashEntry> d_table;
    DevBuf<double> d_mu, d_info;
    DevBuf<int> d_vid;
    unsigned mask = 0;
    bool have_map = false;

    DevScan scan;
    std::vector<PtI> source;
    SourceFilter src_filter;
    bool host_timing = false;  // FLS_HOST_TIMING=1: print the host-side split of every map update
    bool lanes_kernel = true;  // FLS_NDT_LANES=0: one lane per point (ndt_kernel) instead of one lane per neighbour voxel
    bool fused_tail = true;    // the Gauss-Newton tail in the correspondence kernel's last workgroup instead of its own launch (gn_solve_lu_kernel); FLS_FUSED_TAIL=0: separate launch.
                               // Rounds 3-5 measured the fused form SLOWER on configs[2] (83.9 vs 76-80 us per Match) and kept it off: ndt_lanes_kernel<true> held the pose
                               // (sixteen doubles) in registers across the per-point part for the tail -- 144 VGPRs = 3 waves per SIMD = ONE 512-thread workgroup per CU, so
                               // its 457 workgroups ran in two rounds.  Round 6 parks the pose in LDS: 118 VGPRs like the plain kernel, one round, and the launch the tail
                               // no longer needs is a gain: 73.5-74.1 vs 78.3-79.6 us per Match, same pose to the last bit (tools/gpu_ab_ndt.py, profiles/r06_ai_*).
    DevBuf<unsigned> d_ticket;
    DevBuf<int> d_hit_vid;
    DevBuf<unsigned char> d_eff7;
    double final_T[16]{};
    bool have_final = false;
    CellGridImage fitness_grid;
    bool have_fitness_grid = false;

    fls_status init() {
        if (unset_d(p.ndt_voxel_size) || unset_d(p.ndt_res_outlier_threshold) || unset_d(p.rotation_converge_thres) ||
            unset_d(p.position_converge_thres) || unset_f(p.source_cloud_filter_size) || p.ndt_min_points_in_voxel == 0x7fffffff ||
            p.ndt_max_points_in_voxel == 0x7fffffff || p.ndt_min_effective_pts == 0x7fffffff || p.ndt_capacity == 0x7fffffff)
            return FLS_ERR_INVALID;  // CHECK_NE block incremental_ndt.h:27-36
        if (!(p.ndt_voxel_size > 0.0) || !(p.source_cloud_filter_size > 0.f) || p.ndt_capacity <= 0) return FLS_ERR_INVALID;
        init_common();
        src_filter.init();
        if (const char* e = std::getenv("FLS_HOST_TIMING")) host_timing = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_NDT_LANES")) lanes_kernel = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_FUSED_TAIL")) fused_tail = std::atoi(e) != 0;
        d_ticket.reserve(kTicketWords);
        FLS_HIP(hipMemsetAsync(d_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));
        if (const char* e = std::getenv("FLS_NDT_DEVICE_UPDATE")) allow_device_update = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_NDT_DEVICE_MARGIN")) { const long c = std::atol(e); if (c >= 0) device_margin = size_t(c); }
        if (const char* e = std::getenv("FLS_NDT_DEVICE_EVICT")) device_evict = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_NDT_DEVICE_SLACK")) { const long c = std::atol(e); if (c >= 0) device_slack = size_t(c); }
        inv_voxel = 1.0 / p.ndt_voxel_size;
        return FLS_OK;
    }

    static void mean_cov(const std::vector<double>& pts, double* mean, double* cov) {  // ComputeMeanAndCov :91-110
        const double* P = pts.data();
        hm::ndt_mean_cov(int(pts.size() / 3), [P](int k, double* q) { q[0] = P[3 * k]; q[1] = P[3 * k + 1]; q[2] = P[3 * k + 2]; }, mean, cov);
    }
    void update_voxel(Voxel& v) const {  // UpdateVoxel :130-179
        if (flag_first_scan) {
            if (v.pts.size() / 3 > 1u) { mean_cov(v.pts, v.mu, v.sigma); hm::ndt_regularised_info(v.sigma, v.info); }
            else {
                v.mu[0] = v.pts[0]; v.mu[1] = v.pts[1]; v.mu[2] = v.pts[2];
                for (int k = 0; k < 9; ++k) v.info[k] = ((k % 4 == 0) ? 1.0 : 0.0) * 1.0e2;
            }
            v.estimated = true;
            v.dirty = true;
            v.pts.clear();
            return;
        }
        if (v.estimated && v.num_points > p.ndt_max_points_in_voxel) return;
        const int npts = int(v.pts.size() / 3);
        if (!v.estimated && npts > p.ndt_min_points_in_voxel) {
            mean_cov(v.pts, v.mu, v.sigma);
            hm::ndt_regularised_info(v.sigma, v.info);
            v.estimated = true;
            v.dirty = true;
            v.pts.clear();
        } else if (v.estimated && npts > p.ndt_min_points_in_voxel) {
            double cmu[3], cvar[9];
            mean_cov(v.pts, cmu, cvar);
            hm::ndt_merge(v.mu, v.sigma, v.info, v.num_points, cmu, cvar, npts);
            v.num_points += npts;
            v.dirty = true;
            v.pts.clear();
        }
    }

    // ---- device image: table {key -> row}, rows {mu, info, vid}.  Rows are stable per voxel (free list), evicted keys leave a
    // tombstone in the table (never equal to a packed key, never "empty": probing walks over it); one map update ships only
    // the rows and table entries it changed -- a full rebuild when the table must grow, when tombstones pile up, or when
    // most of the image changed anyway (first scan).
    static constexpr unsigned long long kTombKey = ~0ull - 1ull;
    std::vector<int> free_rows;
    unsigned n_rows = 0, n_in_table = 0, n_tomb = 0;
    size_t row_cap = 0;
    std::vector<unsigned> changed_idx;
    std::vector<NdtTableEdit> edit_table;
    std::vector<NdtRowEdit> edit_rows;
    std::vector<unsigned long long> evicted_image_keys;
    PinnedBuf<unsigned char> edit_stage;
    DevBuf<unsigned char> d_edit;
    unsigned long long full_rebuilds = 0, incremental_updates = 0;

    void rebuild_image() {
        size_t n_est = 0;
        for (int v = lru_head; v >= 0; v = pool[v].next) n_est += pool[v].estimated ? 1 : 0;
        const unsigned ts = GridImage::table_size_for(n_est + n_est / 4 + 1024);  // room for the voxels the next scans estimate
        mask = ts - 1;
        h_table.assign(ts, HashEntry{kEmptyKey, 0u, 0u});
        h_mu.clear(); h_info.clear(); h_vid.clear();
        free_rows.clear();
        unsigned slot = 0;
        for (int vi = lru_head; vi >= 0; vi = pool[vi].next) {
            Voxel& v = pool[vi];
            v.dirty = false;
            v.slot = -1;
            if (!v.estimated) continue;
            const unsigned long long key = pack_key(v.kx, v.ky, v.kz);
            unsigned h = hash_key(key) & mask;
            while (h_table[h].key != kEmptyKey) h = (h + 1) & mask;
            h_table[h] = HashEntry{key, slot, 1u};
            h_mu.insert(h_mu.end(), v.mu, v.mu + 3);
            h_info.insert(h_info.end(), v.info, v.info + 9);
            h_vid.push_back(v.vid);
            v.slot = int(slot);
            ++slot;
        }
        n_rows = slot;
        n_in_table = slot;
        n_tomb = 0;
        row_cap = size_t(slot) + slot / 4 + 1024;
        d_table.reserve(ts);
        d_mu.reserve(3 * row_cap);
        d_info.reserve(9 * row_cap);
        d_vid.reserve(row_cap);
        FLS_HIP(hipMemcpyAsync(d_table.p, h_table.data(), ts * sizeof(HashEntry), hipMemcpyHostToDevice, stream));
        if (slot) {
            FLS_HIP(hipMemcpyAsync(d_mu.p, h_mu.data(), h_mu.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            FLS_HIP(hipMemcpyAsync(d_info.p, h_info.data(), h_info.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            FLS_HIP(hipMemcpyAsync(d_vid.p, h_vid.data(), h_vid.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        }
        FLS_HIP(hipStreamSynchronize(stream));
        edit_table.clear(); edit_rows.clear(); evicted_image_keys.clear();
        have_map = true;
        ++full_rebuilds;
    }

    // ships what this call changed; `touched` = pool indices whose UpdateVoxel ran
    void sync_image(const std::vector<int>& touched) {
        if (!have_map) { rebuild_image(); return; }
        edit_table.clear();
        edit_rows.clear();
        changed_idx.clear();
        size_t n_new = 0;
        for (int vi : touched) n_new += (pool[vi].dirty && pool[vi].slot < 0) ? 1 : 0;
        const size_t ts = size_t(mask) + 1;
        if ((size_t(n_in_table) + n_tomb + n_new) * 2 + 2 > ts || size_t(n_rows) + n_new > row_cap || touched.size() * 4 > size_t(n_in_table) + 4096) {
            rebuild_image();
            return;
        }
        for (const unsigned long long key : evicted_image_keys) {  // evictions first: their rows are reused below
            unsigned h = hash_key(key) & mask;
            while (h_table[h].key != key) h = (h + 1) & mask;
            free_rows.push_back(int(h_table[h].begin));
            h_table[h] = HashEntry{kTombKey, 0u, 0u};
            changed_idx.push_back(h);
            --n_in_table;
            ++n_tomb;
        }
        evicted_image_keys.clear();
        for (int vi : touched) {
            Voxel& v = pool[vi];
            if (!v.dirty) continue;
            v.dirty = false;
            if (v.slot < 0) {
                if (!free_rows.empty()) { v.slot = free_rows.back(); free_rows.pop_back(); }
                else v.slot = int(n_rows++);
                const unsigned long long key = pack_key(v.kx, v.ky, v.kz);
                unsigned h = hash_key(key) & mask;
                while (h_table[h].key != kEmptyKey && h_table[h].key != kTombKey) h = (h + 1) & mask;
                if (h_table[h].key == kTombKey) --n_tomb;
                h_table[h] = HashEntry{key, unsigned(v.slot), 1u};
                changed_idx.push_back(h);
                ++n_in_table;
            }
            NdtRowEdit r;
            r.row = unsigned(v.slot);
            r.vid = v.vid;
            std::memcpy(r.mu, v.mu, sizeof(r.mu));
            std::memcpy(r.info, v.info, sizeof(r.info));
            edit_rows.push_back(r);
        }
        ++incremental_updates;
        // one edit per table index, carrying its FINAL entry (a tombstone written and re-used within this call is one edit)
        std::sort(changed_idx.begin(), changed_idx.end());
        changed_idx.erase(std::unique(changed_idx.begin(), changed_idx.end()), changed_idx.end());
        for (const unsigned h : changed_idx) edit_table.push_back(NdtTableEdit{h, 0u, h_table[h]});
        changed_idx.clear();
        if (edit_table.empty() && edit_rows.empty()) return;
        const size_t bt = edit_table.size() * sizeof(NdtTableEdit), br = edit_rows.size() * sizeof(NdtRowEdit);
        edit_stage.reserve(bt + br);
        d_edit.reserve(bt + br);
        if (bt) std::memcpy(edit_stage.p, edit_table.data(), bt);
        if (br) std::memcpy(edit_stage.p + bt, edit_rows.data(), br);
        FLS_HIP(hipMemcpyAsync(d_edit.p, edit_stage.p, bt + br, hipMemcpyHostToDevice, stream));
        const int nt = int(edit_table.size()), nr = int(edit_rows.size());
        hipLaunchKernelGGL(ndt_apply_edits_kernel, dim3(unsigned((std::max(nt, nr * 4) + 255) / 256)), dim3(256), 0, stream,
                           (const NdtTableEdit*)d_edit.p, nt, (const NdtRowEdit*)(d_edit.p + bt), nr, d_table.p, d_mu.p, d_info.p, d_vid.p);
        FLS_HIP(hipStreamSynchronize(stream));  // the staging buffer is reused by the next call
    }

    // ---- device map update (kernels_ndt_update.hpp): the image holds EVERY alive voxel (table count = "estimated"), the
    // per-voxel bookkeeping lives in device rows, the host mirror (pool / LRU list / key map) is stale until
    // sync_host_from_device().  Entered after a host-path update in mapping mode, left for good by the first refused batch.
    bool allow_device_update = true;  // FLS_NDT_DEVICE_UPDATE
    size_t device_margin = 4096;      // FLS_NDT_DEVICE_MARGIN: voxels below the LRU capacity at which device mode is not entered
    size_t device_slack = 65536;      // FLS_NDT_DEVICE_SLACK: spare table entries / rows allocated ahead (test hook: small values force growth)
    bool device_mode = false, device_left = false;
    bool device_evict = true;         // FLS_NDT_DEVICE_EVICT=0: a batch that reaches the LRU capacity is refused (round-2 behaviour; with it the margin rule applies)
    unsigned host_updates_since_left = 0;  // device mode is re-entered after a refusal once eight host-path updates went by (hysteresis)
    unsigned upd_seq = 0;
    size_t dev_entries = 0;           // table entries in use (alive voxels + tombstones since the last re-hash)
    unsigned long long device_evictions = 0, device_compactions = 0;
    DevBuf<unsigned long long> r_key, r_stamp;
    DevBuf<unsigned> r_hslot;
    DevBuf<unsigned long long> r_touch;
    DevBuf<unsigned> u_crank, u_evict, u_sidx, u_srow;  // eviction selection: creation indices, rows in eviction order, the re-created voxels
    unsigned long long device_recreated = 0;            // voxels evicted and re-created inside one device batch
    DevicePairSort ev_sort;
    DevBuf<int> r_np;
    DevBuf<unsigned char> r_est, r_cc;
    DevBuf<double> r_carry, d_sigma;
    DevBuf<NdtUpdState> d_upd;
    PinnedBuf<NdtUpdState> h_upd;
    DevBuf<unsigned> u_slot, u_lx, u_bt;
    DevBuf<float> u_cloud;
    PinnedBuf<float> u_stage;
    DevicePairSort u_sort;
    size_t dev_rows = 0, dev_alive = 0, dev_table = 0;
    int dev_next_vid = 0;
    unsigned long long dev_epoch = 0, device_batches = 0, refused_batches = 0;
    size_t alive() const { return device_mode ? dev_alive : n_alive; }
    NdtRows rows_dev() { return NdtRows{r_key.p, r_hslot.p, r_np.p, r_est.p, r_cc.p, r_carry.p, d_mu.p, d_sigma.p, d_info.p, d_vid.p, r_stamp.p, r_touch.p}; }
    void reserve_rows(size_t cap, bool keep) {
        r_key.reserve(cap, keep, stream); r_stamp.reserve(cap, keep, stream); r_hslot.reserve(cap, keep, stream); r_np.reserve(cap, keep, stream);
        r_touch.reserve(cap, keep, stream, /*zero_new=*/true);
        r_est.reserve(cap, keep, stream); r_cc.reserve(cap, keep, stream); r_carry.reserve(cap * kNdtCarry * 3, keep, stream);
        d_sigma.reserve(cap * 9, keep, stream); d_mu.reserve(cap * 3, keep, stream); d_info.reserve(cap * 9, keep, stream); d_vid.reserve(cap, keep, stream);
        row_cap = cap;
    }
    bool can_enter_device_mode() const {
        // after a refusal (device_left) the handle comes back once eight host-path updates went by: a scan that left the key range or
        // an eviction the device could not order exactly is an episode, not a reason to stay on the 2 ms host path for good
        const bool left = device_left && host_updates_since_left < 8;
        const bool room = device_evict ? true : n_alive + device_margin < size_t(p.ndt_capacity);  // (without device evictions: stay clear of the capacity)
        return allow_device_update && !left && !device_mode && !owner && !p.is_localization_mode && !flag_first_scan &&
               p.ndt_min_points_in_voxel <= kNdtCarry && p.ndt_min_points_in_voxel >= 0 && room && p.ndt_capacity > 2;
    }
    // uploads the whole host mirror as rows (LRU order: row 0 = least recently touched) + a table holding every alive voxel
    void enter_device_mode(size_t batch_hint) {
        const size_t na = n_alive;
        const size_t ahead = device_slack ? na / 2 + 2 * batch_hint + device_slack : 1;  // slack 0 (test hook): nothing ahead, every batch grows
        const unsigned ts = GridImage::table_size_for(na + ahead);
        mask = ts - 1;
        h_table.assign(ts, HashEntry{kEmptyKey, kNdtNewBit, 0u});
        std::vector<unsigned long long> k(na), st(na);
        std::vector<unsigned> hs(na);
        std::vector<int> np(na), vid(na);
        std::vector<unsigned char> est(na), cc(na);
        std::vector<double> carry(na * kNdtCarry * 3, 0.0), mu(na * 3), sg(na * 9), inf(na * 9);
        size_t r = 0;
        for (int vi = lru_tail; vi >= 0; vi = pool[vi].prev, ++r) {
            Voxel& v = pool[vi];
            const unsigned long long key = pack_key(v.kx, v.ky, v.kz);
            unsigned h = hash_key(key) & mask;
            while (h_table[h].key != kEmptyKey) h = (h + 1) & mask;
            h_table[h] = HashEntry{key, unsigned(r), v.estimated ? 1u : 0u};
            k[r] = key; hs[r] = h; st[r] = r + 1; np[r] = v.num_points; vid[r] = v.vid; est[r] = v.estimated ? 1 : 0;
            // unconsumed points: at most min_points of them matter (a saturated voxel's list is never read again)
            const size_t keep = (v.estimated && v.num_points > p.ndt_max_points_in_voxel) ? 0 : v.pts.size() / 3;
            cc[r] = (unsigned char)std::min<size_t>(keep, kNdtCarry);
            for (size_t q = 0; q < size_t(cc[r]) * 3; ++q) carry[r * kNdtCarry * 3 + q] = v.pts[q];
            std::memcpy(&mu[3 * r], v.mu, sizeof(v.mu)); std::memcpy(&sg[9 * r], v.sigma, sizeof(v.sigma)); std::memcpy(&inf[9 * r], v.info, sizeof(v.info));
            v.slot = -1;
            v.dirty = false;
        }
        reserve_rows(na + ahead, false);
        d_table.reserve(ts);
        auto up = [&](void* d, const void* h, size_t bytes) { if (bytes) FLS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream)); };
        up(d_table.p, h_table.data(), ts * sizeof(HashEntry));
        up(r_key.p, k.data(), na * 8); up(r_stamp.p, st.data(), na * 8); up(r_hslot.p, hs.data(), na * 4); up(r_np.p, np.data(), na * 4);
        up(d_vid.p, vid.data(), na * 4); up(r_est.p, est.data(), na); up(r_cc.p, cc.data(), na); up(r_carry.p, carry.data(), carry.size() * 8);
        up(d_mu.p, mu.data(), mu.size() * 8); up(d_sigma.p, sg.data(), sg.size() * 8); up(d_info.p, inf.data(), inf.size() * 8);
        FLS_HIP(hipStreamSynchronize(stream));
        FLS_HIP(hipMemsetAsync(r_touch.p, 0, r_touch.cap * sizeof(unsigned long long), stream));
        dev_rows = dev_alive = dev_entries = na;
        device_left = false;
        dev_table = ts;
        dev_next_vid = next_vid;
        dev_epoch = na + 1;
        d_upd.reserve(1);
        h_upd.reserve(2);
        edit_table.clear(); edit_rows.clear(); evicted_image_keys.clear(); changed_idx.clear(); free_rows.clear();
        have_map = true;
        device_mode = true;
        ++full_rebuilds;
    }
    // the table of a device-mode image is rebuilt on the device when the next batch might fill it past one half
    unsigned long long table_growths = 0, row_growths = 0;
    void grow_table_device(size_t need_entries) {
        ++table_growths;
        const unsigned ts = GridImage::table_size_for(device_slack ? need_entries + need_entries / 2 + device_slack : need_entries);
        DevBuf<HashEntry> nt;
        nt.reserve(ts);
        hipLaunchKernelGGL(ndt_table_fill_kernel, dim3((ts + 255) / 256), dim3(256), 0, stream, nt.p, ts);
        hipLaunchKernelGGL(ndt_table_rehash_kernel, dim3(unsigned((dev_rows + 255) / 256)), dim3(256), 0, stream, nt.p, ts - 1, rows_dev(), unsigned(dev_rows));
        FLS_HIP(hipStreamSynchronize(stream));
        std::swap(d_table.p, nt.p);
        std::swap(d_table.cap, nt.cap);
        mask = ts - 1;
        dev_table = ts;
        dev_entries = dev_alive;  // the re-hash dropped the tombstones
    }
    // one map update on the device; false = refused (nothing changed): the caller syncs the host mirror and replays on the host
    bool device_add_cloud(const std::vector<PtI>& cloud) {
        const size_t n = cloud.size();
        if (n == 0) return true;
        u_stage.reserve(3 * n);
        u_cloud.reserve(3 * n);
        for (size_t i = 0; i < n; ++i) { u_stage.p[i] = cloud[i].x; u_stage.p[n + i] = cloud[i].y; u_stage.p[2 * n + i] = cloud[i].z; }
        FLS_HIP(hipMemcpyAsync(u_cloud.p, u_stage.p, 3 * n * sizeof(float), hipMemcpyHostToDevice, stream));
        return device_add_cloud_dev(u_cloud.p, u_cloud.p + n, u_cloud.p + 2 * n, n);
    }
    // the filtered world cloud already on the device (x, y, z of n points, cloud order)
    bool device_add_cloud_dev(const float* x, const float* y, const float* z, const size_t n) {
        if (n == 0) return true;
        if (n > size_t(kVgMaxBlocks) * kVgTile) return false;
        if ((dev_entries + n) * 2 + 2 > dev_table) grow_table_device(dev_alive + n);
        if (dev_rows + n > row_cap) { reserve_rows(device_slack ? dev_rows + dev_rows / 2 + 2 * n : dev_rows + n, true); ++row_growths; }
        upd_seq = upd_seq + 1u;
        if (upd_seq == 0u) {  // the touch words order batches by their sequence number: start over after 2^32 of them
            upd_seq = 1u;
            FLS_HIP(hipMemsetAsync(r_touch.p, 0, r_touch.cap * sizeof(unsigned long long), stream));
        }
        const bool may_evict = device_evict && dev_alive + n >= size_t(p.ndt_capacity);  // every point a new voxel: the worst case
        NdtUpdState& hs = h_upd.p[0];
        hs = NdtUpdState{};
        hs.n_rows = unsigned(dev_rows); hs.n_alive = unsigned(dev_alive); hs.next_vid = dev_next_vid; hs.epoch = dev_epoch;
        hs.capacity = unsigned(p.ndt_capacity); hs.row_cap = unsigned(std::min<size_t>(row_cap, 0x7fffffffu));
        hs.seq = upd_seq; hs.evict_ready = may_evict ? 1u : 0u; hs.dead_hi = unsigned((dev_epoch + n + 2) >> 32) + 1u;
        FLS_HIP(hipMemcpyAsync(d_upd.p, &hs, sizeof(NdtUpdState), hipMemcpyHostToDevice, stream));
        const int ni = int(n), nb1 = (ni + 255) / 256, nb2 = (ni + kVgScanBlock - 1) / kVgScanBlock;
        u_slot.reserve(n); u_lx.reserve(n); u_bt.reserve(size_t(2 * nb2));
        const NdtRows R = rows_dev();
        hipLaunchKernelGGL(ndt_upd_locate, dim3(unsigned(nb1)), dim3(256), 0, stream, x, y, z, ni, inv_voxel, d_table.p, mask, u_slot.p, d_upd.p, r_touch.p, upd_seq);
        hipLaunchKernelGGL(ndt_upd_creators, dim3(unsigned(nb2)), dim3(kVgScanBlock), 0, stream, ni, (const HashEntry*)d_table.p, (const unsigned*)u_slot.p, u_lx.p, u_bt.p);
        hipLaunchKernelGGL(vg_scan, dim3(1), dim3(kVgScanBlock), 0, stream, (const unsigned*)u_bt.p, u_bt.p + nb2, nb2, &d_upd.p->n_new);
        hipLaunchKernelGGL(ndt_upd_predecide, dim3(1), dim3(1), 0, stream, d_upd.p);
        if (may_evict && dev_rows > 0) {
            // rows in LRU order: stable radix sort by the low, then by the high word of the 64-bit stamps (dead rows last)
            const unsigned nr = unsigned(dev_rows), nbr = (nr + 255u) / 256u;
            ev_sort.prepare(dev_rows);
            hipLaunchKernelGGL(ndt_evict_keys, dim3(nbr), dim3(256), 0, stream, R, nr, hs.dead_hi, 0, (const unsigned*)nullptr, ev_sort.k0, ev_sort.v0);
            ev_sort.run(4, stream);
            hipLaunchKernelGGL(ndt_evict_keys, dim3(nbr), dim3(256), 0, stream, R, nr, hs.dead_hi, 1, (const unsigned*)ev_sort.v0, ev_sort.k0, ev_sort.v0);
            ev_sort.run(DevicePairSort::passes_for((unsigned long long)hs.dead_hi), stream);
            // the walk over the LRU order (ndt_evict_select): skips what the batch touched in time, re-creates what it touched too late
            u_crank.reserve(n); u_evict.reserve(std::max<size_t>(dev_rows, n)); u_sidx.reserve(size_t(kNdtMaxRecreate)); u_srow.reserve(size_t(kNdtMaxRecreate));
            hipLaunchKernelGGL(ndt_upd_cranks, dim3(unsigned(nb1)), dim3(256), 0, stream, ni, (const unsigned*)u_lx.p, (const unsigned*)(u_bt.p + nb2), u_crank.p);
            hipLaunchKernelGGL(ndt_evict_select, dim3(1), dim3(kNdtEvBlock), 0, stream, R, (const unsigned*)ev_sort.v0, nr, d_upd.p, (const unsigned*)u_crank.p, u_evict.p,
                               u_sidx.p, u_srow.p);
        }
        hipLaunchKernelGGL(ndt_upd_decide, dim3(1), dim3(1), 0, stream, d_upd.p);
        if (may_evict && dev_rows > 0)
            hipLaunchKernelGGL(ndt_evict_apply, dim3(unsigned(nb1)), dim3(256), 0, stream, R, (const unsigned*)u_evict.p, d_table.p, (const NdtUpdState*)d_upd.p);
        hipLaunchKernelGGL(ndt_upd_create, dim3(unsigned(nb1)), dim3(256), 0, stream, x, y, z, ni, inv_voxel, d_table.p, (const unsigned*)u_slot.p,
                           (const unsigned*)u_lx.p, (const unsigned*)(u_bt.p + nb2), R, (const NdtUpdState*)d_upd.p, (const unsigned*)u_crank.p, (const unsigned*)u_sidx.p);
        u_sort.prepare(n);
        hipLaunchKernelGGL(ndt_upd_rowkeys, dim3(unsigned(nb1)), dim3(256), 0, stream, ni, (const HashEntry*)d_table.p, (const unsigned*)u_slot.p, u_sort.k0, u_sort.v0,
                           (const NdtUpdState*)d_upd.p);
        u_sort.run(DevicePairSort::passes_for((unsigned long long)(dev_rows + n)), stream);
        hipLaunchKernelGGL(ndt_upd_voxel, dim3(unsigned((ni + 63) / 64)), dim3(64), 0, stream, (const unsigned*)u_sort.k0, (const unsigned*)u_sort.v0, ni, x, y, z, R,
                           d_table.p, d_upd.p, int(p.ndt_min_points_in_voxel), int(p.ndt_max_points_in_voxel));
        hipLaunchKernelGGL(ndt_upd_commit, dim3(1), dim3(1), 0, stream, d_upd.p, unsigned(n));
        FLS_HIP(hipMemcpyAsync(&h_upd.p[1], d_upd.p, sizeof(NdtUpdState), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        FLS_HIP(hipGetLastError());
        const NdtUpdState& o = h_upd.p[1];
        if (!o.apply) { ++refused_batches; return false; }
        dev_entries += size_t(o.n_rows) - dev_rows - o.recreated;  // one table entry per created voxel (an evicted voxel's entry stays as a tombstone, a re-created voxel reuses its own)
        dev_rows = o.n_rows; dev_alive = o.n_alive; dev_next_vid = o.next_vid; dev_epoch = o.epoch;
        last_touched = o.touched;
        device_evictions += o.evict;
        device_recreated += o.recreated;
        ++device_batches;
        // retired rows pile up at the capacity (hundreds per scan): drop them through the host mirror now and then
        if (dev_rows - dev_alive > std::max<size_t>(2 * dev_alive, 262144)) compact_device_rows(n);
        return true;
    }
    unsigned last_touched = 0;
    // opt-in device source filter (FLS_DEVICE_VOXELGRID=1): the filtered scan never left the device -- transform it, filter it
    // again (the reference's second VoxelGrid, :186) and update the map without touching the host.  false: not applied (the
    // caller takes the host path, which also reports a key out of range)
    DevBuf<float> w_cloud;
    DeviceVoxelGrid map_vg;
    unsigned long long resident_updates = 0;
    bool device_update_from_resident_source(const double* T_in) {
        const size_t n = src_filter.vg.n_out;
        if (n == 0) return false;
        const auto t0 = std::chrono::steady_clock::now();
        w_cloud.reserve(3 * n);
        XformF xf;
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) xf.R[i + j * 3] = float(T_in[i + j * 4]);
        for (int i = 0; i < 3; ++i) xf.t[i] = float(T_in[12 + i]);
        hipLaunchKernelGGL(xform_cloud_f_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, src_filter.vg.ox(), src_filter.vg.oy(), src_filter.vg.oz(),
                           int(n), xf, w_cloud.p, w_cloud.p + n, w_cloud.p + 2 * n);
        if (!map_vg.run(w_cloud.p, w_cloud.p + n, w_cloud.p + 2 * n, src_filter.vg.oi(), n, p.source_cloud_filter_size, stream)) return false;
        if (!device_add_cloud_dev(map_vg.ox(), map_vg.oy(), map_vg.oz(), map_vg.n_out)) {
            sync_host_from_device();
            return false;
        }
        ++resident_updates;
        if (host_timing) {
            const auto t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[fls_reg] ndt map update (device, resident source): %zu -> %zu pts | transform + VoxelGrid + UpdateVoxel x%u %.3f ms (%zu voxels)\n", n,
                         map_vg.n_out, last_touched, std::chrono::duration<double, std::milli>(t1 - t0).count(), dev_alive);
        }
        return true;
    }
    // device rows -> host mirror (pool, LRU list in stamp order, key map); the image is rebuilt by the next host-path update
    void sync_host_from_device() {
        const size_t nr = dev_rows;
        std::vector<unsigned long long> k(nr), st(nr);
        std::vector<int> np(nr), vid(nr);
        std::vector<unsigned char> est(nr), cc(nr);
        std::vector<double> carry(nr * kNdtCarry * 3), mu(nr * 3), sg(nr * 9), inf(nr * 9);
        auto dn = [&](void* h, const void* d, size_t bytes) { if (bytes) FLS_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream)); };
        dn(k.data(), r_key.p, nr * 8); dn(st.data(), r_stamp.p, nr * 8); dn(np.data(), r_np.p, nr * 4); dn(vid.data(), d_vid.p, nr * 4);
        dn(est.data(), r_est.p, nr); dn(cc.data(), r_cc.p, nr); dn(carry.data(), r_carry.p, carry.size() * 8);
        dn(mu.data(), d_mu.p, mu.size() * 8); dn(sg.data(), d_sigma.p, sg.size() * 8); dn(inf.data(), d_info.p, inf.size() * 8);
        FLS_HIP(hipStreamSynchronize(stream));
        std::vector<size_t> order(nr);
        for (size_t i = 0; i < nr; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return st[a] < st[b]; });  // stamps are unique
        pool.clear(); free_ix.clear(); grids.clear();
        lru_head = lru_tail = -1;
        n_alive = 0;
        pool.reserve(nr);
        for (const size_t r : order) {  // oldest first: every push_front leaves the most recent at the head
            if (k[r] == kNdtDeadKey) continue;  // evicted on the device
            pool.emplace_back();
            Voxel& v = pool.back();
            unpack_key(k[r], v.kx, v.ky, v.kz);
            v.pts.assign(carry.begin() + r * kNdtCarry * 3, carry.begin() + r * kNdtCarry * 3 + size_t(cc[r]) * 3);
            std::memcpy(v.mu, &mu[3 * r], sizeof(v.mu)); std::memcpy(v.sigma, &sg[9 * r], sizeof(v.sigma)); std::memcpy(v.info, &inf[9 * r], sizeof(v.info));
            v.estimated = est[r] != 0; v.num_points = np[r]; v.vid = vid[r];
            const int vi = int(pool.size()) - 1;
            lru_push_front(vi);
            grids.insert(k[r], vi);
            ++n_alive;
        }
        next_vid = dev_next_vid;
        device_mode = false;
        device_left = true;
        host_updates_since_left = 0;
        have_map = false;  // the host-path image (estimated voxels only, host-assigned rows) is rebuilt from the mirror
        rebuild_image();
    }
    void compact_device_rows(size_t batch_hint) {
        sync_host_from_device();
        device_left = false;
        ++device_compactions;
        if (can_enter_device_mode()) enter_device_mode(batch_hint);
    }

    fls_status add_cloud_impl(const std::vector<PtI>& cloud_world_full) {  // :182-227
        const auto t0 = std::chrono::steady_clock::now();
        const std::vector<PtI> cloud_world = voxel_grid(cloud_world_full, p.source_cloud_filter_size);
        const auto t1 = std::chrono::steady_clock::now();
        // range check first (all-or-nothing)
        for (const PtI& pt : cloud_world) {
            const double f[3] = {double(pt.x) * inv_voxel, double(pt.y) * inv_voxel, double(pt.z) * inv_voxel};
            for (int a = 0; a < 3; ++a)
                if (!(std::fabs(f[a]) < double(kKeyLimit))) return FLS_ERR_RANGE;
        }
        if (p.is_localization_mode) { have_fitness_grid = fitness_grid.build(cloud_world, 1.0f, stream) == FLS_OK; }
        if (device_mode) {
            if (!flag_first_scan && device_add_cloud(cloud_world)) {
                if (host_timing) {
                    const auto t4 = std::chrono::steady_clock::now();
                    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                    std::fprintf(stderr, "[fls_reg] ndt map update (device): %zu -> %zu pts | VoxelGrid %.3f ms | upload + device UpdateVoxel x%u %.3f ms (%zu voxels)\n",
                                 cloud_world_full.size(), cloud_world.size(), ms(t0, t1), last_touched, ms(t1, t4), dev_alive);
                }
                return FLS_OK;
            }
            sync_host_from_device();  // refused (a batch that would evict, a key out of range): exact sequential replay below
        }
        ++epoch;
        std::vector<int> touched;
        touched.reserve(4096);
        for (const PtI& pt : cloud_world) {
            const double pe[3] = {double(pt.x), double(pt.y), double(pt.z)};
            const int kx = int(pe[0] * inv_voxel), ky = int(pe[1] * inv_voxel), kz = int(pe[2] * inv_voxel);  // cast<int>: truncation (:195)
            const unsigned long long key = pack_key(kx, ky, kz);
            int vi = grids.find(key);
            if (vi < 0) {
                if (!free_ix.empty()) { vi = free_ix.back(); free_ix.pop_back(); }
                else { vi = int(pool.size()); pool.emplace_back(); }
                Voxel& v = pool[vi];
                v.kx = kx; v.ky = ky; v.kz = kz;
                v.pts.clear();
                v.pts.push_back(pe[0]); v.pts.push_back(pe[1]); v.pts.push_back(pe[2]);
                v.estimated = false;
                v.num_points = 1;
                v.vid = next_vid++;
                v.slot = -1;
                v.dirty = false;
                v.epoch = 0;
                lru_push_front(vi);
                grids.insert(key, vi);
                ++n_alive;
                if (n_alive >= size_t(p.ndt_capacity)) {  // :202-205
                    const int b = lru_tail;
                    Voxel& e = pool[b];
                    const unsigned long long ek = pack_key(e.kx, e.ky, e.kz);
                    grids.erase(ek);
                    if (e.slot >= 0) evicted_image_keys.push_back(ek);
                    e.slot = -1;
                    e.epoch = 0;  // a voxel evicted within the call that touched it gets no UpdateVoxel (the reference would dereference a null entry)
                    lru_unlink(b);
                    free_ix.push_back(b);
                    --n_alive;
                    if (b == vi) continue;  // capacity 1: the new voxel itself
                }
            } else {
                Voxel& v = pool[vi];
                v.pts.push_back(pe[0]); v.pts.push_back(pe[1]); v.pts.push_back(pe[2]);
                if (!v.estimated) v.num_points++;
                if (lru_head != vi) { lru_unlink(vi); lru_push_front(vi); }
            }
            Voxel& v = pool[vi];
            if (v.epoch != epoch) { v.epoch = epoch; touched.push_back(vi); }
        }
        const auto t2 = std::chrono::steady_clock::now();
        size_t w = 0;
        for (size_t k = 0; k < touched.size(); ++k) {
            const int vi = touched[k];
            if (pool[vi].epoch != epoch) continue;  // evicted after its touch
            pool[vi].epoch = epoch + 0x40000000u;   // a pool entry re-created after an eviction appears twice: processed once
            update_voxel(pool[vi]);
            touched[w++] = vi;
        }
        touched.resize(w);
        for (int vi : touched) pool[vi].epoch = epoch;
        flag_first_scan = p.is_localization_mode ? true : false;  // :222-226
        const auto t3 = std::chrono::steady_clock::now();
        sync_image(touched);
        if (device_left) ++host_updates_since_left;
        if (can_enter_device_mode()) enter_device_mode(cloud_world.size());
        if (host_timing) {
            const auto t4 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[fls_reg] ndt map update: %zu -> %zu pts | VoxelGrid %.3f ms | insert + LRU %.3f ms | UpdateVoxel x%zu %.3f ms | image sync %.3f ms (%zu voxels)\n",
                         cloud_world_full.size(), cloud_world.size(), ms(t0, t1), ms(t1, t2), touched.size(), ms(t2, t3), ms(t3, t4), n_alive);
        }
        return FLS_OK;
    }
    fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) override {
        if (c1 != nullptr && n1 != 0) return FLS_ERR_INVALID;
        return add_cloud_impl(cloud_from(c0, n0, stride));
    }
    fls_status scan_upload(const float* s0, size_t n0, const float*, size_t, int stride) override {
        src_filter.filter(s0, n0, stride, p.source_cloud_filter_size, stream, scan, source);  // :232
        return FLS_OK;
    }
    fls_status scan_upload_raw(const float* s0, size_t n0, const float*, size_t, int stride) override {
        src_filter.upload_raw_only(s0, n0, stride, p.source_cloud_filter_size, stream, scan, source);  // (no filtered scan is resident until the next Match: fitness answers FLS_ERR_STATE)
        return FLS_OK;
    }
    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        const NdtMatcher& M = owner ? *owner : *this;  // a batch lane reads its owner's voxel tables
        if (M.alive() == 0 || !M.have_map) return FLS_ERR_STATE;  // CHECK(!grids_.empty()) :230
        if (src_filter.raw_pending) src_filter.refilter(stream, scan, source);  // :231-232, on the resident raw scan
        const size_t n = scan.n;
        const int nblk = int((n + 63) / 64);
        stats = fls_stats{};
        stats.n_source = int(n);
        double T_in[16];
        std::memcpy(T_in, T, sizeof(T_in));
        d_hit_vid.reserve(std::max<size_t>(n * 7, 1));
        d_eff7.reserve(std::max<size_t>(n * 7, 1));
        d_partials_b.reserve(size_t(std::max(nblk, 1)) * kPartialStride);
        const NdtGridDev ng{M.d_table.p, M.mask, M.d_mu.p, M.d_info.p, M.d_vid.p, M.inv_voxel};
        Pose16 T0;
        std::memcpy(T0.m, T, sizeof(T0.m));
        const unsigned word = run_mailbox_loop(int(p.max_iterations), n, [&](int it, int first) {
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it], stream));
            if (nblk > 0 && lanes_kernel && !count_traffic) {  // one lane per neighbour voxel (64 points per workgroup: same row count)
                const LuTailArgs tail{1, p.rotation_converge_thres, p.position_converge_thres, p.ndt_min_effective_pts, mb_dev, launch_word()};
                if (fused_tail)
                    hipLaunchKernelGGL(ndt_lanes_kernel<true>, dim3(nblk), dim3(kNdtLanesBlock), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first,
                                       T0, ng, p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, d_ticket.p, 8, tail);
                else
                    hipLaunchKernelGGL(ndt_lanes_kernel<false>, dim3(nblk), dim3(kNdtLanesBlock), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first,
                                       T0, ng, p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, (unsigned*)nullptr, 8, tail);
                if (fused_tail) {  // the last workgroup ran the Gauss-Newton tail: one launch per iteration
                    if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
                    return;
                }
            } else if (nblk > 0) {
                if (count_traffic)
                    hipLaunchKernelGGL(ndt_kernel<true>, dim3(nblk), dim3(64), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first, T0, ng,
                                       p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, d_tc.p);
                else
                    hipLaunchKernelGGL(ndt_kernel<false>, dim3(nblk), dim3(64), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first, T0, ng,
                                       p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, d_tc.p);
            }
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
            hipLaunchKernelGGL(gn_solve_lu_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)d_partials_b.p, nblk, 1,
                               p.rotation_converge_thres, p.position_converge_thres, p.ndt_min_effective_pts, mb_dev, launch_word());
        });
        const Mailbox& s = *mb_host;
        stats.iterations = int(word & 0xffu);
        stats.n_valid = s.n_valid;
        stats.sum_res = s.sum_res;
        std::memcpy(stats.last_dx, s.last_dx, sizeof(stats.last_dx));
        // the min_effective early-out sets done with converged == 0 on the device (:306-309: T = pose; return false)
        const bool early_fail = s.done && !s.converged && s.n_valid < p.ndt_min_effective_pts;
        if (early_fail) {
            std::memcpy(T, s.T, sizeof(double) * 16);
            stats.converged = 0;
            if (out) *out = stats;
            return FLS_NOT_CONVERGED;
        }
        // has_converge = true unconditionally (:325, Q10)
        fls_status rc = FLS_OK;
        if (!p.is_localization_mode && update_map && !owner) {
            // Q11: the cloud is transformed with the INPUT T (:327-329)
            if (device_mode && !flag_first_scan && src_filter.resident && device_update_from_resident_source(T_in)) {
                stats.map_updated = 1;
            } else {
                src_filter.materialize(stream, source);
                const fls_status arc = add_cloud_impl(hm::xform_cloud_f(source, T_in));
                if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
            }
        }
        std::memcpy(T, s.T, sizeof(double) * 16);
        std::memcpy(final_T, s.T, sizeof(final_T));
        have_final = true;
        stats.converged = 1;
        if (out) *out = stats;
        return rc;
    }
    const NdtMatcher* owner = nullptr;
    std::unique_ptr<fls_matcher> clone_for_lane() override {
        auto q = std::make_unique<NdtMatcher>();
        q->kind = kind; q->p = p; q->device = device;
        if (q->init() != FLS_OK) return nullptr;
        q->owner = this;
        return q;
    }
    fls_status prepare_batch() override { FLS_HIP(hipStreamSynchronize(stream)); return have_map ? FLS_OK : FLS_ERR_STATE; }
    fls_status fitness(float max_range, float* score) override {
        if (!p.is_localization_mode) { *score = std::numeric_limits<float>::max(); return FLS_OK; }  // :346-348
        if (!have_fitness_grid || src_filter.withdrawn) return FLS_ERR_STATE;
        // A Match that stops at the effective-point floor returns before `final_transformation_ = T` (incremental_ndt.h:306-309 vs :335): the score is then
        // taken with the PREVIOUS Match's transformation and the new source cloud -- and, when no Match of this matcher has got that far yet, with
        // the value-initialised member `final_transformation_{}` (:395), the zero matrix in the compiled reference: every point lands on the origin.
        // Found by running the localization-mode random scenarios on the GPU (round 6, `lfuzz3`: FLS_ERR_STATE where the reference answers).
        static const double zero_T[16] = {0.0};
        return fitness_score_device(*this, fitness_grid, scan, have_final ? final_T : zero_T, max_range, score);
    }
    int correspondences(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) override {
        const size_t n = std::min(cap, scan.n);
        if (!n) return 0;
        std::vector<unsigned char> ef(n * 7);
        FLS_HIP(hipMemcpyAsync(ids, d_hit_vid.p, n * 7 * sizeof(int), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipMemcpyAsync(ef.data(), d_eff7.p, n * 7, hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < n; ++i) {
            int c = 0;
            for (int k = 0; k < 7; ++k) c += ef[i * 7 + k];
            cnt[i] = uint8_t(c);
            valid[i] = c > 0;
        }
        return int(n);
    }
    size_t map_size(int slot) const override {
        if (slot == 105) return size_t(src_filter.device_runs);
        if (slot == 106) return size_t(src_filter.host_runs);
        if (slot == 107) return size_t(full_rebuilds);  // image: full rebuilds / incremental updates
        if (slot == 108) return size_t(incremental_updates);
        if (slot == 109) return size_t(device_batches);  // map updates applied on the device / refused (replayed on the host)
        if (slot == 110) return size_t(refused_batches);
        if (slot == 111) return size_t(resident_updates);  // map updates fed by the device-resident filtered scan
        if (slot == 112) return size_t(table_growths);     // device-side table rebuilds / row-array growths
        if (slot == 113) return size_t(row_growths);
        if (slot == 126) return size_t(device_recreated);    // ... of which re-created by a later point of the same batch
        if (slot == 117) return size_t(device_evictions);    // voxels evicted by device batches / row compactions through the host mirror
        if (slot == 118) return size_t(device_compactions);
        return alive();
    }
};

}  // namespace fls
