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
        inv_voxel = 1.0 / p.ndt_voxel_size;
        return FLS_OK;
    }

    static void mean_cov(const std::vector<double>& pts, double* mean, double* cov) {  // ComputeMeanAndCov :91-110
        const size_t len = pts.size() / 3;
        double s[3] = {0, 0, 0};
        for (size_t k = 0; k < len; ++k) { s[0] += pts[3 * k]; s[1] += pts[3 * k + 1]; s[2] += pts[3 * k + 2]; }
        for (int a = 0; a < 3; ++a) mean[a] = s[a] / double(len);
        double c[9] = {0};
        for (size_t k = 0; k < len; ++k) {
            const double v[3] = {pts[3 * k] - mean[0], pts[3 * k + 1] - mean[1], pts[3 * k + 2] - mean[2]};
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) c[i + j * 3] += v[i] * v[j];
        }
        for (int k = 0; k < 9; ++k) cov[k] = c[k] / double(len - 1);
    }
    void regularised_info(Voxel& v) const {
        double m[9];
        for (int k = 0; k < 9; ++k) m[k] = v.sigma[k] + ((k % 4 == 0) ? 1.0 : 0.0) * 1.0e-3;
        hm::inv3(m, v.info);
    }
    void update_voxel(Voxel& v) const {  // UpdateVoxel :130-179
        if (flag_first_scan) {
            if (v.pts.size() / 3 > 1u) { mean_cov(v.pts, v.mu, v.sigma); regularised_info(v); }
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
            regularised_info(v);
            v.estimated = true;
            v.dirty = true;
            v.pts.clear();
        } else if (v.estimated && npts > p.ndt_min_points_in_voxel) {
            double cmu[3], cvar[9], nmu[3], nvar[9];
            mean_cov(v.pts, cmu, cvar);
            const int hm_ = v.num_points, cn = npts;  // UpdateMeanAndCov :112-120
            for (int a = 0; a < 3; ++a) nmu[a] = (double(hm_) * v.mu[a] + double(cn) * cmu[a]) / double(hm_ + cn);
            const double dh[3] = {v.mu[0] - nmu[0], v.mu[1] - nmu[1], v.mu[2] - nmu[2]};
            const double dc[3] = {cmu[0] - nmu[0], cmu[1] - nmu[1], cmu[2] - nmu[2]};
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i)
                    nvar[i + j * 3] = (double(hm_) * (v.sigma[i + j * 3] + dh[i] * dh[j]) + double(cn) * (cvar[i + j * 3] + dc[i] * dc[j])) / double(hm_ + cn);
            std::memcpy(v.mu, nmu, sizeof(nmu));
            std::memcpy(v.sigma, nvar, sizeof(nvar));
            v.num_points += npts;
            v.dirty = true;
            v.pts.clear();
            double U[9], S[3], V[9];
            hm::svd3(v.sigma, U, S, V);
            if (S[1] < S[0] * 1e-3) S[1] = S[0] * 1e-3;
            if (S[2] < S[0] * 1e-3) S[2] = S[0] * 1e-3;
            const double il[3] = {1.0 / S[0], 1.0 / S[1], 1.0 / S[2]};
            double VL[9];
            for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) VL[i + k * 3] = V[i + k * 3] * il[k];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i)
                    v.info[i + j * 3] = (VL[i] * U[j] + VL[i + 3] * U[j + 3]) + VL[i + 6] * U[j + 6];
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
    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        const NdtMatcher& M = owner ? *owner : *this;  // a batch lane reads its owner's voxel tables
        if (M.n_alive == 0 || !M.have_map) return FLS_ERR_STATE;  // CHECK(!grids_.empty()) :230
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
            if (nblk > 0) {
                if (count_traffic)
                    hipLaunchKernelGGL(ndt_kernel<true>, dim3(nblk), dim3(64), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first, T0, ng,
                                       p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, d_tc.p);
                else
                    hipLaunchKernelGGL(ndt_kernel<false>, dim3(nblk), dim3(64), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first, T0, ng,
                                       p.ndt_res_outlier_threshold, d_hit_vid.p, d_eff7.p, d_partials_b.p, d_tc.p);
            }
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
            hipLaunchKernelGGL(gn_solve_lu_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)d_partials_b.p, nblk, 1,
                               p.rotation_converge_thres, p.position_converge_thres, p.ndt_min_effective_pts, mb_dev, match_id);
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
            src_filter.materialize(stream, source);
            const fls_status arc = add_cloud_impl(hm::xform_cloud_f(source, T_in));  // Q11: transformed with the INPUT T (:327-329)
            if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
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
        if (!have_fitness_grid || !have_final) return FLS_ERR_STATE;
        return fitness_score_device(*this, fitness_grid, scan, final_T, max_range, score);
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
        return n_alive;
    }
};

}  // namespace fls
