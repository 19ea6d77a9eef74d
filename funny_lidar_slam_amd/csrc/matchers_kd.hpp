// matchers_kd.hpp -- host side of the three kinds whose reference uses pcl::KdTreeFLANN:
//   IcpMatcher       <- IcpOptimized<double>            include/registration/icp_optimized.h
//   LoamFullMatcher  <- LoamFull<double>                include/registration/loam_full_kdtree.h
//   P2PlaneKdMatcher <- LoamPointToPlaneKdtree<double>  include/registration/loam_point_to_plane_kdtree.h
// Map bookkeeping (deque of clouds, VoxelGrid, rebuild) stays on the host like the reference's
// AddCloudToLocalMap; the kd-tree is replaced by an exact-kNN hash grid rebuilt at the same moments.
#pragma once
#include "matcher_base.hpp"
#include "host_math.hpp"
#include "device_voxelgrid.hpp"
#include "kernels_knn.hpp"
#include "kernels_grid_coop.hpp"
#include "fitness_host.hpp"
#include <deque>

namespace fls {

// LOAM feature maps: cells of half the gate radius, 5x5x5 block searched in two stages (kernels_grid_coop.hpp): the inner
// 27 cells hold 4-8x fewer candidates than 27 gate-sized cells and finish almost every query (-10 % kernel time; the
// kernel is bound by the 27 hash probes, not by the candidates)
constexpr int kGridRings = 2;
inline float cell_for_gate(double gate_sq) {  // smallest safe cell for a squared-distance gate
    return float(std::sqrt(gate_sq) * 1.0001);
}

// ---------------------------------------------------------------------------------------------
struct IcpMatcher final : fls_matcher {
    std::deque<std::vector<PtI>> cloud_deque;
    std::vector<PtI> local_map, source;
    SourceFilter src_filter;
    CellGridImage grid;
    KdMapDevice mapdev;      // cell-grid build on the device (default); deque + map-side VoxelGrid on the device (opt-in)
    DeviceCloudRing ring;    // the deque's clouds back to back on the device (opt-in path only)
    size_t local_map_n = 0;  // points of the local map (the host vector is not produced on the device path)
    bool have_map = false;
    bool fused = true;  // FLS_ICP_FUSED=0: separate correspondence and fit launches
    bool fused_tail = true;  // FLS_FUSED_TAIL=0: gn_solve_lu_kernel as its own launch
    DevBuf<unsigned> d_ticket;
    const IcpMatcher* owner = nullptr;  // batch lane: reads the owner's map grid
    DevScan scan;
    size_t raw_n = 0;
    hm::KeyframeGate gate;
    double final_T[16]{};
    bool have_final = false;
    DevBuf<int> d_nn_id;
    DevBuf<unsigned char> d_eff, d_nn_cnt;
    DevBuf<float4> d_nn_pts;
    DevBuf<float> d_kth;

    fls_status init() {
        if (unset_f(p.map_cloud_filter_size) || unset_f(p.source_cloud_filter_size) || unset_d(p.point_search_thres) ||
            unset_d(p.position_converge_thres) || unset_d(p.rotation_converge_thres) || unset_d(p.rot_thre_add_cloud) ||
            unset_d(p.dist_thre_add_cloud) || p.local_map_size == 0x7fffffffu)
            return FLS_ERR_INVALID;  // CHECK_NE block icp_optimized.h:33-41
        if (!(p.point_search_thres > 0.0) || !(p.map_cloud_filter_size > 0.f) || !(p.source_cloud_filter_size > 0.f)) return FLS_ERR_INVALID;
        init_common();
        src_filter.init();
        mapdev.init();
        if (const char* e = std::getenv("FLS_ICP_FUSED")) fused = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_FUSED_TAIL")) fused_tail = std::atoi(e) != 0;
        d_ticket.reserve(kTicketWords);
        FLS_HIP(hipMemsetAsync(d_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));
        return FLS_OK;
    }
    fls_status add_cloud_impl(const std::vector<PtI>& new_cloud) {  // :165-189
        const bool dev = mapdev.vg_on_device && mapdev.grid_on_device;
        if (p.is_localization_mode) {
            cloud_deque.clear();
            cloud_deque.push_back(new_cloud);
            if (dev) { ring.clear(); ring.push_back(new_cloud, stream); }
        } else {
            cloud_deque.push_back(new_cloud);
            if (dev) ring.push_back(new_cloud, stream);
            if (cloud_deque.size() > p.local_map_size) { cloud_deque.pop_front(); if (dev) ring.pop_front(); }
        }
        // gate-sized cells here: the ICP scan starts far from the map (1-NN often beyond half the gate in the early
        // iterations), the two-stage search would run its second stage for most queries (measured 25 vs 13.5 us / launch)
        const float cell = cell_for_gate(p.point_search_thres);
        // opt-in device path: the deque is resident, VoxelGrid (Q13: always) + grid build never leave the device
        if (dev && mapdev.filter_and_build(grid, ring, true, p.map_cloud_filter_size, cell, 1, false, local_map_n, stream)) {
            local_map.clear();
            have_map = true;
            return FLS_OK;
        }
        local_map.clear();
        for (const auto& c : cloud_deque) local_map.insert(local_map.end(), c.begin(), c.end());
        local_map = voxel_grid(local_map, p.map_cloud_filter_size);  // Q13: always
        ++mapdev.host_filters;
        local_map_n = local_map.size();
        const fls_status rc = mapdev.build_from_host(grid, local_map, cell, stream);
        have_map = rc == FLS_OK;
        return rc;
    }
    fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) override {
        if (c1 != nullptr && n1 != 0) return FLS_ERR_INVALID;
        return add_cloud_impl(cloud_from(c0, n0, stride));
    }
    fls_status scan_upload(const float* s0, size_t n0, const float*, size_t, int stride) override {
        raw_n = n0;
        src_filter.filter(s0, n0, stride, p.source_cloud_filter_size, stream, scan, source);  // :57
        return FLS_OK;
    }
    fls_status scan_upload_raw(const float* s0, size_t n0, const float*, size_t, int stride) override {
        raw_n = n0;
        src_filter.upload_raw_only(s0, n0, stride, p.source_cloud_filter_size, stream, scan, source);
        have_final = false;  // (no filtered scan is resident until the next Match: fls_get_fitness_score answers FLS_ERR_STATE)
        return FLS_OK;
    }
    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        if (raw_n <= 10) return FLS_ERR_INVALID;  // CHECK_GT(ordered_cloud_.size(), 10u) :55
        if (src_filter.raw_pending) src_filter.refilter(stream, scan, source);  // :57, on the resident raw scan
        if (!(owner ? owner->have_map : have_map)) return FLS_ERR_STATE;
        const size_t n = scan.n;
        const int nwg = int((n + 255) / 256);
        stats = fls_stats{};
        stats.n_source = int(n);
        d_nn_pts.reserve(std::max<size_t>(n, 1));
        d_nn_cnt.reserve(std::max<size_t>(n, 1));
        d_kth.reserve(std::max<size_t>(n, 1));
        d_nn_id.reserve(std::max<size_t>(n, 1));
        d_eff.reserve(std::max<size_t>(n, 1));
        const CellGridDev cg = cell_dev(owner ? owner->grid : grid);
        const dim3 knn_grid_dim(unsigned((((n * 8 + 255) / 256) + 63) / 64 * 64));  // multiple of 64: the XCD chunk re-map is a bijection
        d_partials_b.reserve(size_t(std::max<unsigned>(knn_grid_dim.x, unsigned(std::max(nwg, 1)))) * kPartialStride);
        Pose16 T0;
        std::memcpy(T0.m, T, sizeof(T0.m));
        const unsigned word = run_mailbox_loop(int(p.max_iterations), n, [&](int it, int first) {
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it], stream));
            if (fused) {  // search + fit in one launch: one partial row per workgroup of the search grid
                const LuTailArgs tail{0, p.rotation_converge_thres, p.position_converge_thres, 0, mb_dev, launch_word()};
                hipLaunchKernelGGL(icp_knn_fit_kernel, knn_grid_dim, dim3(256), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first,
                                   T0, cg, float(p.point_search_thres), p.point_search_thres, d_nn_id.p, d_eff.p, d_partials_b.p,
                                   fused_tail ? d_ticket.p : (unsigned*)nullptr, 8, tail);
                if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
                if (fused_tail) return;  // ... and the Gauss-Newton tail in its last workgroup: one launch per iteration
                hipLaunchKernelGGL(gn_solve_lu_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)d_partials_b.p,
                                   int(knn_grid_dim.x), 0, p.rotation_converge_thres, p.position_converge_thres, 0, mb_dev, launch_word());
                return;
            }
            hipLaunchKernelGGL((grid_knn_kernel<1, true>), knn_grid_dim, dim3(256), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p,
                               first, T0, cg, float(p.point_search_thres), d_nn_pts.p, d_nn_cnt.p, d_kth.p, (unsigned char*)nullptr);
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
            hipLaunchKernelGGL(icp_fit_kernel, dim3(nwg), dim3(256), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_state.p, first, T0,
                               (const float4*)d_nn_pts.p, (const unsigned char*)d_nn_cnt.p, (const float*)d_kth.p, p.point_search_thres,
                               d_nn_id.p, d_eff.p, d_partials_b.p);
            hipLaunchKernelGGL(gn_solve_lu_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)d_partials_b.p,
                               nwg, 0, p.rotation_converge_thres, p.position_converge_thres, 0, mb_dev, launch_word());
        });
        const Mailbox& mb = *mb_host;
        std::memcpy(T, mb.T, sizeof(double) * 16);
        std::memcpy(final_T, mb.T, sizeof(final_T));
        have_final = true;
        const bool has_converge = mb.converged != 0;  // Q10: false after max iterations
        stats.iterations = int(word & 0xffu);
        stats.n_valid = mb.n_valid;
        stats.sum_res = mb.sum_res;
        std::memcpy(stats.last_dx, mb.last_dx, sizeof(stats.last_dx));
        stats.converged = has_converge ? 1 : 0;
        fls_status rc = has_converge ? FLS_OK : FLS_NOT_CONVERGED;
        // :153 `has_converge_ && IsNeedAddCloud(T) && !is_localization_mode_`: the gate (and its last_T) is evaluated before the mode
        // switch, as in the reference; update_map == 0 (this ABI's "registration only" switch) skips the whole statement,
        // so such a call leaves the keyframe gate alone
        if (update_map && !owner && has_converge && gate.need(final_T, p.dist_thre_add_cloud, p.rot_thre_add_cloud) && !p.is_localization_mode) {
            src_filter.materialize(stream, source);
            const fls_status arc = add_cloud_impl(hm::xform_cloud_f(source, final_T));
            if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
        }
        if (out) *out = stats;
        return rc;
    }
    void reset_job_state() override { gate = hm::KeyframeGate(); have_final = false; }  // function-static last_T of a fresh process (Q12)
    std::unique_ptr<fls_matcher> clone_for_lane() override {
        auto q = std::make_unique<IcpMatcher>();
        q->kind = kind; q->p = p; q->device = device;
        if (q->init() != FLS_OK) return nullptr;
        q->owner = this;
        return q;
    }
    fls_status prepare_batch() override { FLS_HIP(hipStreamSynchronize(stream)); return have_map ? FLS_OK : FLS_ERR_STATE; }
    fls_status fitness(float max_range, float* score) override {
        if (owner || !have_map || !have_final) return FLS_ERR_STATE;
        return fitness_score_device(*this, grid, scan, final_T, max_range, score);
    }
    int correspondences(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) override {
        const size_t n = std::min(cap, scan.n);
        if (!n) return 0;
        std::vector<int> id(n);
        std::vector<unsigned char> ef(n);
        FLS_HIP(hipMemcpyAsync(id.data(), d_nn_id.p, n * sizeof(int), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipMemcpyAsync(ef.data(), d_eff.p, n, hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < n; ++i) { ids[i] = id[i]; cnt[i] = id[i] >= 0 ? 1 : 0; valid[i] = ef[i]; }
        return int(n);
    }
    size_t map_size(int slot) const override {
        if (slot == 105) return size_t(src_filter.device_runs);  // source filters run on the device / on the host
        if (slot == 106) return size_t(src_filter.host_runs);
        if (slot == 114) return size_t(mapdev.builder.builds);    // cell grids built on the device / map updates filtered on the device /
        if (slot == 115) return size_t(mapdev.device_filters);    // ... filtered on the host
        if (slot == 116) return size_t(mapdev.host_filters);
        return local_map_n;
    }
};

// ---------------------------------------------------------------------------------------------
// per-feature-class device buffers of the LOAM kinds
struct FeatureDev {
    DevScan scan;
    DevBuf<float4> nn_pts;           // [n][5] neighbours left by grid_knn_kernel<5>
    DevBuf<unsigned char> nn_cnt, flag, cnt_out;
    DevBuf<float> kth;               // d2 of the 5th neighbour
    DevBuf<int> nn_id;               // [n][5] reported ids (gate-accepted sets only)
    DevBuf<double> J;
    void prepare() {
        const size_t n = std::max<size_t>(scan.n, 1);
        nn_pts.reserve(n * 5); nn_cnt.reserve(n); cnt_out.reserve(n); kth.reserve(n); nn_id.reserve(n * 5); flag.reserve(n); J.reserve(7 * n);
    }
    int fetch(hipStream_t s, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) {
        const size_t n = std::min(cap, scan.n);
        if (!n) return 0;
        FLS_HIP(hipMemcpyAsync(ids, nn_id.p, n * 5 * sizeof(int), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipMemcpyAsync(cnt, cnt_out.p, n, hipMemcpyDeviceToHost, s));
        FLS_HIP(hipMemcpyAsync(valid, flag.p, n, hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        return int(n);
    }
    GridKnnArgs knn_args(const CellGridDev& cg, float gate) {
        return GridKnnArgs{scan.x.p, scan.y.p, scan.z.p, int(scan.n), cg, gate, nn_pts.p, nn_cnt.p, kth.p, flag.p};
    }
    FeatureFitArgs fit_args(float gate, double thres, double* partials) {
        return FeatureFitArgs{scan.x.p, scan.y.p, scan.z.p, int(scan.n), nn_pts.p, nn_cnt.p, kth.p, gate, thres, nn_id.p, cnt_out.p, J.p, flag.p, partials};
    }
    // grid_knn_kernel<5> (flags cleared by the first iteration: once per Match, Q1) + feature_fit_kernel<LINE>
    template <bool LINE>
    void launch(hipStream_t s, GnState* st, int first, const Pose16& T0, const CellGridDev& cg, float gate, double thres, double* partials) {
        const size_t n = scan.n;
        if (n == 0) return;
        if (cg.rings == 1 && cg.by_id != nullptr) {  // gate-sized cells + the cloud by index: the one-stage 27-cell kernel
            const dim3 g27(unsigned((((n * 4 + 255) / 256) + 63) / 64 * 64));
            hipLaunchKernelGGL((grid_knn27_kernel<false>), g27, dim3(256), 0, s, scan.x.p, scan.y.p, scan.z.p, int(n), st, first, T0, cg, nn_pts.p, nn_cnt.p,
                               kth.p, flag.p);
        } else {
            const dim3 knn_grid_dim(unsigned((((n * 8 + 255) / 256) + 63) / 64 * 64));  // multiple of 64: the XCD chunk re-map is a bijection
            if (std::isinf(gate))  // un-gated search (LoamPointToPlaneKdtree): the instantiation with the ring walk
                hipLaunchKernelGGL((grid_knn_kernel<5, false, true>), knn_grid_dim, dim3(256), 0, s, scan.x.p, scan.y.p, scan.z.p, int(n), st, first, T0, cg, gate,
                                   nn_pts.p, nn_cnt.p, kth.p, flag.p);
            else
                hipLaunchKernelGGL((grid_knn_kernel<5, false>), knn_grid_dim, dim3(256), 0, s, scan.x.p, scan.y.p, scan.z.p, int(n), st, first, T0, cg, gate,
                                   nn_pts.p, nn_cnt.p, kth.p, flag.p);
        }
        hipLaunchKernelGGL((feature_fit_kernel<LINE>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, scan.x.p, scan.y.p, scan.z.p, int(n), st,
                           first, T0, (const float4*)nn_pts.p, (const unsigned char*)nn_cnt.p, (const float*)kth.p, gate, thres, nn_id.p,
                           cnt_out.p, J.p, flag.p, partials);
    }
};

struct LoamFullMatcher final : fls_matcher {
    std::deque<std::vector<PtI>> corner_deque, planar_deque;
    std::vector<PtI> local_corner, local_planar;
    CellGridImage corner_grid, planar_grid;
    KdMapDevice mapdev_planar, mapdev_corner;
    DeviceCloudRing ring_planar, ring_corner;
    size_t local_planar_n = 0, local_corner_n = 0;
    bool have_map = false;
    const LoamFullMatcher* owner = nullptr;  // batch lane: reads the owner's two map grids
    FeatureDev corner, planar;
    hm::KeyframeGate gate;
    bool grid27 = false;     // FLS_GRID27=1: gate-sized cells + the one-stage 27-cell kernel (measured slower: A/B switch)
    bool dual_launch = true;  // FLS_LOAM_DUAL=0: one correspondence + one fit launch per feature class
    bool loam_fused_tail = true;  // FLS_FUSED_TAIL=0: gn_solve_loam_kernel as a launch of its own after the dual fit launch
    DevBuf<unsigned> d_loam_ticket;

    fls_status init() {
        if (const char* e = std::getenv("FLS_GRID27")) grid27 = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_LOAM_DUAL")) dual_launch = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_FUSED_TAIL")) loam_fused_tail = std::atoi(e) != 0;
        if (unset_d(p.point_to_planar_thres) || unset_d(p.point_search_thres) || unset_d(p.line_ratio_thres) ||
            unset_d(p.position_converge_thres) || unset_d(p.rotation_converge_thres) || unset_d(p.rot_thre_add_cloud) ||
            unset_d(p.dist_thre_add_cloud))
            return FLS_ERR_INVALID;  // CHECK_NE block loam_full_kdtree.h:44-53
        if (p.local_planar_size == 0 || p.local_corner_size == 0) return FLS_ERR_INVALID;  // CHECK_GT :55-56
        if (!(p.point_search_thres > 0.0) || !(p.corner_voxel_filter_size > 0.f) || !(p.planar_voxel_filter_size > 0.f)) return FLS_ERR_INVALID;
        init_common();
        d_loam_ticket.reserve(kTicketWords);
        FLS_HIP(hipMemsetAsync(d_loam_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));
        mapdev_planar.init();
        mapdev_corner.init();
        return FLS_OK;
    }
    fls_status add_cloud_impl(const std::vector<PtI>& planar_cloud, const std::vector<PtI>& corner_cloud) {  // :65-104
        const bool dev = mapdev_planar.vg_on_device && mapdev_planar.grid_on_device;
        corner_deque.push_back(corner_cloud);
        planar_deque.push_back(planar_cloud);
        if (dev) { ring_corner.push_back(corner_cloud, stream); ring_planar.push_back(planar_cloud, stream); }
        if (planar_deque.size() > p.local_planar_size) { planar_deque.pop_front(); if (dev) ring_planar.pop_front(); }
        if (corner_deque.size() > p.local_corner_size) { corner_deque.pop_front(); if (dev) ring_corner.pop_front(); }
        // FLS_GRID27=1: gate-sized cells + the one-stage 27-cell kernel; default: half-gate cells + the two-stage kernel
        const bool g27 = grid27;
        const float cs = (g27 ? 1.0f : 0.5f) * cell_for_gate(p.point_search_thres);
        const int rings = g27 ? 1 : kGridRings;
        // one feature class: [VoxelGrid of] the concatenated deque (the filter only once the deque holds more than 5 frames, :92-100)
        auto rebuild = [&](std::deque<std::vector<PtI>>& dq, DeviceCloudRing& ring, KdMapDevice& md, CellGridImage& grid, std::vector<PtI>& local, size_t& n_local,
                           float leaf) -> fls_status {
            const bool filter = dq.size() > 5;
            if (dev && md.filter_and_build(grid, ring, filter, leaf, cs, rings, g27, n_local, stream)) { local.clear(); return FLS_OK; }
            local.clear();
            for (const auto& c : dq) local.insert(local.end(), c.begin(), c.end());
            if (filter) { local = voxel_grid(local, leaf); ++md.host_filters; }
            n_local = local.size();
            return md.build_from_host(grid, local, cs, stream, rings, g27);
        };
        fls_status rc = rebuild(planar_deque, ring_planar, mapdev_planar, planar_grid, local_planar, local_planar_n, p.planar_voxel_filter_size);
        if (rc != FLS_OK) return rc;
        rc = rebuild(corner_deque, ring_corner, mapdev_corner, corner_grid, local_corner, local_corner_n, p.corner_voxel_filter_size);
        have_map = rc == FLS_OK;
        return rc;
    }
    fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) override {
        if (c1 == nullptr && n1 != 0) return FLS_ERR_INVALID;  // CHECK_EQ(cloud_list.size(), 2) :66
        return add_cloud_impl(cloud_from(c0, n0, stride), c1 ? cloud_from(c1, n1, stride) : std::vector<PtI>());
    }
    fls_status scan_upload(const float* s0, size_t n0, const float* s1, size_t n1, int stride) override {
        planar.scan.upload(cloud_from(s0, n0, stride), stream);
        corner.scan.upload(s1 ? cloud_from(s1, n1, stride) : std::vector<PtI>(), stream);
        return FLS_OK;
    }
    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        if (!(owner ? owner->have_map : have_map)) return FLS_ERR_STATE;
        const size_t np = planar.scan.n, nc = corner.scan.n;
        const int nbp = int((np + 255) / 256), nbc = int((nc + 255) / 256);
        stats = fls_stats{};
        stats.n_source = int(np);
        stats.n_source_corner = int(nc);
        planar.prepare();
        corner.prepare();
        d_partials_a.reserve(size_t(std::max(nbc, 1)) * kPartialStride);
        d_partials_b.reserve(size_t(std::max(nbp, 1)) * kPartialStride);
        const CellGridDev cgp = cell_dev(owner ? owner->planar_grid : planar_grid), cgc = cell_dev(owner ? owner->corner_grid : corner_grid);
        const float gate_f = float(p.point_search_thres);
        Pose16 T0;
        std::memcpy(T0.m, T, sizeof(T0.m));
        const unsigned word = run_mailbox_loop(int(p.max_iterations), np + nc, [&](int it, int first) {
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it], stream));
            if (dual_launch && nc != 0 && np != 0 && !(cgc.rings == 1 && cgc.by_id != nullptr)) {
                // both classes in one correspondence launch and one fit launch (they are independent until the solve)
                const int kc = int((((nc * 8 + 255) / 256) + 63) / 64 * 64), kp = int((((np * 8 + 255) / 256) + 63) / 64 * 64);
                hipLaunchKernelGGL((grid_knn_dual_kernel<5, false>), dim3(unsigned(kc + kp)), dim3(256), 0, stream, (const GnState*)d_state.p, first, T0,
                                   corner.knn_args(cgc, gate_f), planar.knn_args(cgp, gate_f), kc);
                const bool fuse = loam_fused_tail;
                const LoamFusedTail tail{(const double*)d_partials_a.p, (const double*)d_partials_b.p, nbc, nbp, p.rotation_converge_thres, p.position_converge_thres,
                                         fuse ? d_loam_ticket.p : nullptr, 8, mb_dev, launch_word()};
                if (fuse)
                    hipLaunchKernelGGL(feature_fit_dual_kernel<true>, dim3(unsigned(nbc + nbp)), dim3(256), 0, stream, (const GnState*)d_state.p, first, T0,
                                       corner.fit_args(gate_f, p.line_ratio_thres, d_partials_a.p), planar.fit_args(gate_f, p.point_to_planar_thres, d_partials_b.p), nbc, tail);
                else
                    hipLaunchKernelGGL(feature_fit_dual_kernel<false>, dim3(unsigned(nbc + nbp)), dim3(256), 0, stream, (const GnState*)d_state.p, first, T0,
                                       corner.fit_args(gate_f, p.line_ratio_thres, d_partials_a.p), planar.fit_args(gate_f, p.point_to_planar_thres, d_partials_b.p), nbc, tail);
                if (fuse) {
                    if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
                    return;
                }
            } else {
                corner.launch<true>(stream, d_state.p, first, T0, cgc, gate_f, p.line_ratio_thres, d_partials_a.p);
                planar.launch<false>(stream, d_state.p, first, T0, cgp, gate_f, p.point_to_planar_thres, d_partials_b.p);
            }
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
            hipLaunchKernelGGL(gn_solve_loam_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)d_partials_a.p,
                               nbc, (const double*)d_partials_b.p, nbp, p.rotation_converge_thres, p.position_converge_thres, mb_dev, launch_word());
        });
        const Mailbox& mb = *mb_host;
        std::memcpy(T, mb.T, sizeof(double) * 16);
        bool has_converge = true;
        if (mb.n_valid < 50) has_converge = false;  // number_valid_planar_ < 50 :181
        stats.iterations = int(word & 0xffu);
        stats.n_valid = mb.n_valid;
        stats.n_valid_corner = mb.n_valid2;
        stats.sum_res = mb.sum_res;
        stats.sum_res_corner = mb.sum_res2;
        std::memcpy(stats.last_dx, mb.last_dx, sizeof(stats.last_dx));
        stats.converged = has_converge ? 1 : 0;
        fls_status rc = has_converge ? FLS_OK : FLS_NOT_CONVERGED;
        if (update_map && !owner && has_converge && gate.need(mb.T, p.dist_thre_add_cloud, p.rot_thre_add_cloud)) {  // :185-193 (no localization switch)
            const fls_status arc = add_cloud_impl(hm::xform_cloud_d(planar.scan.host, mb.T), hm::xform_cloud_d(corner.scan.host, mb.T));
            if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
        }
        if (out) *out = stats;
        return rc;
    }
    void reset_job_state() override { gate = hm::KeyframeGate(); }  // function-static last_T of a fresh process (Q12)
    std::unique_ptr<fls_matcher> clone_for_lane() override {
        auto q = std::make_unique<LoamFullMatcher>();
        q->kind = kind; q->p = p; q->device = device;
        if (q->init() != FLS_OK) return nullptr;
        q->owner = this;
        return q;
    }
    fls_status prepare_batch() override { FLS_HIP(hipStreamSynchronize(stream)); return have_map ? FLS_OK : FLS_ERR_STATE; }
    fls_status fitness(float, float* score) override { *score = std::numeric_limits<float>::max(); return FLS_OK; }  // FloatNaN :206-208
    int correspondences(int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) override {
        return (slot == 1 ? corner : planar).fetch(stream, ids, cnt, valid, cap);
    }
    size_t map_size(int slot) const override {
        if (slot == 114) return size_t(mapdev_planar.builder.builds + mapdev_corner.builder.builds);
        if (slot == 115) return size_t(mapdev_planar.device_filters + mapdev_corner.device_filters);
        if (slot == 116) return size_t(mapdev_planar.host_filters + mapdev_corner.host_filters);
        return slot == 1 ? local_corner_n : local_planar_n;
    }
};

// ---------------------------------------------------------------------------------------------
struct P2PlaneKdMatcher final : fls_matcher {
    std::deque<std::vector<PtI>> cloud_deque;
    std::vector<PtI> local_map;
    CellGridImage grid;
    KdMapDevice mapdev;
    DeviceCloudRing ring;
    size_t local_map_n = 0;
    bool have_map = false;
    const P2PlaneKdMatcher* owner = nullptr;  // batch lane: reads the owner's map grid
    FeatureDev planar;
    hm::KeyframeGate gate;
    double final_T[16]{};
    bool have_final = false;

    fls_status init() {
        if (unset_d(p.point_to_planar_thres) || unset_d(p.position_converge_thres) || unset_d(p.rotation_converge_thres) ||
            unset_d(p.rot_thre_add_cloud) || unset_d(p.dist_thre_add_cloud) || unset_f(p.map_cloud_filter_size))
            return FLS_ERR_INVALID;  // CHECK_NE block loam_point_to_plane_kdtree.h:43-50
        if (!(p.map_cloud_filter_size > 0.f)) return FLS_ERR_INVALID;
        init_common();
        mapdev.init();
        return FLS_OK;
    }
    fls_status add_cloud_impl(const std::vector<PtI>& planar_cloud) {  // :56-79
        const bool dev = mapdev.vg_on_device && mapdev.grid_on_device;
        if (p.is_localization_mode) {
            cloud_deque.clear();
            cloud_deque.push_back(planar_cloud);
            if (dev) { ring.clear(); ring.push_back(planar_cloud, stream); }
        } else {
            cloud_deque.push_back(planar_cloud);
            if (dev) ring.push_back(planar_cloud, stream);
            if (cloud_deque.size() > p.local_map_size) { cloud_deque.pop_front(); if (dev) ring.pop_front(); }
        }
        // un-gated 5-NN: ring search with a cell of two map leaves (>= 1 point per leaf after VoxelGrid)
        const float cell = std::max(2.0f * p.map_cloud_filter_size, 0.5f);
        if (dev && mapdev.filter_and_build(grid, ring, true, p.map_cloud_filter_size, cell, 1, false, local_map_n, stream)) {
            local_map.clear();
            have_map = true;
            return FLS_OK;
        }
        local_map.clear();
        for (const auto& c : cloud_deque) local_map.insert(local_map.end(), c.begin(), c.end());
        local_map = voxel_grid(local_map, p.map_cloud_filter_size);
        ++mapdev.host_filters;
        local_map_n = local_map.size();
        const fls_status rc = mapdev.build_from_host(grid, local_map, cell, stream);
        have_map = rc == FLS_OK;
        return rc;
    }
    fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) override {
        if (c1 != nullptr && n1 != 0) return FLS_ERR_INVALID;
        return add_cloud_impl(cloud_from(c0, n0, stride));
    }
    fls_status scan_upload(const float* s0, size_t n0, const float*, size_t, int stride) override {
        planar.scan.upload(cloud_from(s0, n0, stride), stream);
        return FLS_OK;
    }
    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        if (!(owner ? owner->have_map : have_map)) return FLS_ERR_STATE;
        const size_t n = planar.scan.n;
        const int nblk = int((n + 255) / 256);
        stats = fls_stats{};
        stats.n_source = int(n);
        planar.prepare();
        d_partials_b.reserve(size_t(std::max(nblk, 1)) * kPartialStride);
        const CellGridDev cg = cell_dev(owner ? owner->grid : grid);
        Pose16 T0;
        std::memcpy(T0.m, T, sizeof(T0.m));
        const unsigned word = run_mailbox_loop(int(p.max_iterations), n, [&](int it, int first) {
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it], stream));
            planar.launch<false>(stream, d_state.p, first, T0, cg, INFINITY, p.point_to_planar_thres, d_partials_b.p);  // un-gated (:219)
            if (profiling) FLS_HIP(hipEventRecord(ev[2 * it + 1], stream));
            hipLaunchKernelGGL(gn_solve_loam_kernel, dim3(1), dim3(kSolveThreads), 0, stream, d_state.p, first, T0, (const double*)nullptr, 0,
                               (const double*)d_partials_b.p, nblk, p.rotation_converge_thres, p.position_converge_thres, mb_dev, launch_word());
        });
        const Mailbox& mb = *mb_host;
        std::memcpy(T, mb.T, sizeof(double) * 16);
        std::memcpy(final_T, mb.T, sizeof(final_T));
        have_final = true;
        bool has_converge = true;
        if (mb.n_valid < 50) has_converge = false;
        stats.iterations = int(word & 0xffu);
        stats.n_valid = mb.n_valid;
        stats.sum_res = mb.sum_res;
        std::memcpy(stats.last_dx, mb.last_dx, sizeof(stats.last_dx));
        stats.converged = has_converge ? 1 : 0;
        fls_status rc = has_converge ? FLS_OK : FLS_NOT_CONVERGED;
        if (update_map && !owner && has_converge && gate.need(final_T, p.dist_thre_add_cloud, p.rot_thre_add_cloud) && !p.is_localization_mode) {  // :145-149
            const fls_status arc = add_cloud_impl(hm::xform_cloud_f(planar.scan.host, final_T));
            if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
        }
        if (out) *out = stats;
        return rc;
    }
    void reset_job_state() override { gate = hm::KeyframeGate(); have_final = false; }
    std::unique_ptr<fls_matcher> clone_for_lane() override {
        auto q = std::make_unique<P2PlaneKdMatcher>();
        q->kind = kind; q->p = p; q->device = device;
        if (q->init() != FLS_OK) return nullptr;
        q->owner = this;
        return q;
    }
    fls_status prepare_batch() override { FLS_HIP(hipStreamSynchronize(stream)); return have_map ? FLS_OK : FLS_ERR_STATE; }
    fls_status fitness(float max_range, float* score) override {
        if (owner || !have_map || !have_final) return FLS_ERR_STATE;
        return fitness_score_device(*this, grid, planar.scan, final_T, max_range, score);
    }
    int correspondences(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) override { return planar.fetch(stream, ids, cnt, valid, cap); }
    size_t map_size(int slot) const override {
        if (slot == 114) return size_t(mapdev.builder.builds);
        if (slot == 115) return size_t(mapdev.device_filters);
        if (slot == 116) return size_t(mapdev.host_filters);
        return local_map_n;
    }
};

}  // namespace fls
