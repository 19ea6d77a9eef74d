// matcher_p2plane_ivox.hpp -- host side of FLS_P2PLANE_IVOX, the replacement of
// LoamPointToPlaneIVOX<double> (include/registration/loam_point_to_plane_ivox.h:30-355).
//
//   AddCloudToLocalMap  :60-139   first call / localization: insert all; later: the
//                                 centre-distance down-sampling rule on the last kNN result
//   Match               :141-216  device-resident Gauss-Newton loop (2 launches / iteration, the host waits on a
//                                 mailbox word instead of synchronising the stream)
//   GetFitnessScore     :225-253  localization mode only (FloatNaN otherwise)
#pragma once
#include "matcher_base.hpp"
#include "kernels_p2plane.hpp"
#include "kernels_knn.hpp"
#include "kernels_ivox_coop.hpp"
#include "kernels_ivox_update.hpp"
#include "ivox_image.hpp"
#include "fitness_host.hpp"
#include "device_voxelgrid.hpp"
#include <thread>
#include <chrono>
#include <hip/hip_ext.h>

namespace fls {

struct P2PlaneIvoxMatcher final : fls_matcher {
    HostIvox ivox;
    IvoxImage image;
    bool image_dirty = true, image_built = false;
    bool rebuild_after_replay = false;  // the device refused a batch for lack of room: the next refresh re-flattens
    size_t n_incremental = 0, n_full_rebuilds = 0;
    // a batch lane (fls_match_batch) reads its owner's resident map image
    const IvoxImage* borrowed = nullptr;
    // a member of a replica set (fls_replicas_*): `image` is a device-to-device copy of the owner's (kNN side only), the host mirror is
    // empty -- the handle serves fls_match_batch; everything that needs the mirror answers FLS_ERR_STATE
    bool replica_only = false;
    bool host_timing = false;  // FLS_HOST_TIMING=1: print the host-side cost of every map update
    // ---- device-side AddPoints (kernels_ivox_update.hpp): while `device_map` is set the DEVICE image is the authoritative map
    // (points, voxel table, LRU stamps, counters) and the host mirror `ivox` is stale; sync_host_from_device() brings it back.
    bool device_map = false;
    bool allow_device_map = true;  // FLS_IVOX_DEVICE_UPDATE=0: always the host path (A/B)
    size_t device_margin = 4096;   // voxels of head-room below the LRU capacity required to (re-)enter device mode (FLS_IVOX_DEVICE_MARGIN: test hook)
    size_t n_device_updates = 0, n_host_fallbacks = 0, n_device_evictions = 0, n_device_recreated = 0, n_refused_conflict = 0, n_refused_full = 0, n_refused_outside = 0;
    bool fused_update = false;     // FLS_IVOX_FUSED_UPDATE=1 (A/B): small batches as ONE launch of one workgroup
    bool short_chain_update = true;  // FLS_IVOX_SHORT_CHAIN=0 (A/B): the round-3 chain of eleven launches for every batch
    size_t n_fused_updates = 0, n_short_updates = 0;
    // the decision + update chain queued behind the iterations the Match is expected to need, gated on the device (ivox_add_decide_kernel):
    // removes the host's mailbox turnaround + first-launch latency (~14 us) in front of the map update.  FLS_IVOX_SPECULATIVE=0: wait first.
    bool speculative_update = true, spec_pending = false, last_chain_skipped = false;
    size_t n_speculative = 0, n_speculative_skipped = 0;
    bool fused_commit = false;     // FLS_IVOX_FUSED_COMMIT=1 (A/B): the last finishing block publishes instead of a commit launch (measured slower: 15-26 us against 4.6 + 4.1 us)
    int finish_blocks = kFinishBlocks;  // FLS_IVOX_FINISH_BLOCKS (A/B)
    bool device_evict = true;      // FLS_IVOX_DEVICE_EVICT=0: a batch that reaches the LRU capacity is refused (round-2 behaviour, with the margin rule)
    DevicePairSort ev_sort;
    DevBuf<unsigned> d_ev_bt, d_crank, d_evict_list;
    DevBuf<IvoxUpdState> d_upd_state;
    IvoxUpdMailbox* upd_mb_host = nullptr;
    IvoxUpdMailbox* upd_mb_dev = nullptr;
    unsigned upd_seq = 0;
    size_t dev_n_points = 0, dev_n_alive = 0, dev_n_bricks = 0;  // mirrored from the update mailbox
    unsigned long long stamp_bound = 0;        // upper bound of every LRU stamp on the device (sizes the second sort round)
    DevBuf<uint2> d_lx, d_bt;
    DevBuf<unsigned> d_seq_src, d_seq_cell, d_jj, d_tlist;
    DevBuf<uint4> d_px, d_bt2;
    DevBuf<unsigned char> d_fbit;
    PinnedBuf<char> upd_stage;
    DevBuf<unsigned char> d_code;
    DevBuf<float4> d_pw;
    std::vector<unsigned char> h_code;
    std::vector<Pt4> h_pw;
    DevBuf<unsigned> d_ticket;
    bool use_dense = true; // dense voxel window instead of the hash table when the map extent allows (FLS_IVOX_DENSE=0 disables)
    int xcd_chunk = 8;     // workgroups per XCD chunk of the kNN block re-map (FLS_IVOX_XCD_CHUNK)
    bool balanced = true;  // equal candidate ranges per lane through an LDS voxel table (FLS_IVOX_BALANCED=0: whole voxels per lane)
    int variant = 4;       // lanes cooperating on one query in ivox_knn_kernel: 4 or 8 (FLS_IVOX_VARIANT)
    int ticket_shards = 8; // fan-in of the fit kernel's workgroups: 8 per-XCD counters + a top counter (FLS_TICKET_SHARDS=1: one counter)
    int fit_threads_env = 0;    // FLS_FIT_THREADS = 512 | 256 (A/B): workgroup size of the fit kernel; 0 = by scan size
    bool plain_launch = false;  // FLS_PLAIN_LAUNCH=1 (A/B): hipLaunchKernelGGL instead of hipExtLaunchKernelGGL with null events
    bool prof_fit = false; // FLS_PROF_FIT=1 (diagnosis): the profiling events bracket the fit+solve kernel instead of the kNN kernel
    bool is_first = true;  // the reference's function-static flag (:62), per handle here
    const double filter_size_map_min = 0.5;  // :351

    DevScan scan;
    // per-point state that outlives an iteration (Q1) or a Match (nearest_points_, :257)
    DevBuf<float4> d_nn;          // [n][5]
    DevBuf<unsigned char> d_nn_cnt;   // neighbour count (bits 0-2) | 0x80 when the point's list is in rows form
    DevBuf<unsigned> d_nn_ids;        // [n][8] map slots of the neighbours (ids form: what the kNN kernel writes by default)
    bool nn_ids_mode = true;          // FLS_IVOX_NN_IDS=0: the kNN kernel writes gathered rows (round-2 behaviour)
    bool nn_rows_current = true;      // every list is in rows form
    size_t nn_n = 0;              // logical size of nearest_points_
    int nn_prev = 0;              // its size before the Match in flight
    DevBuf<double> d_J;           // [7][n]
    DevBuf<unsigned char> d_flag;
    size_t number_planar_point = 0;
    double T_[16]{}, final_T[16]{};
    bool have_final = false;
    // localization-mode fitness map (kd-tree stand-in)
    CellGridImage fitness_grid;
    bool have_fitness_grid = false;
    std::vector<Pt4> h_nn;
    std::vector<unsigned char> h_cnt, h_flag;

    ~P2PlaneIvoxMatcher() override {
        if (stream) (void)hipStreamSynchronize(stream);
        if (upd_mb_host) (void)hipHostFree(upd_mb_host);
    }
    fls_status init() {
        if (unset_d(p.point_to_planar_thres) || unset_d(p.position_converge_thres) || unset_d(p.rotation_converge_thres))
            return FLS_ERR_INVALID;  // CHECK_NE(..., max()) at :45-48
        init_common();
        if (const char* e = std::getenv("FLS_IVOX_VARIANT")) { const int v = std::atoi(e); if (v == 4 || v == 8) variant = v; }
        if (const char* e = std::getenv("FLS_IVOX_DENSE")) use_dense = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_PROF_FIT")) prof_fit = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_PLAIN_LAUNCH")) plain_launch = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_FIT_THREADS")) { const int v = std::atoi(e); fit_threads_env = (v == 256 || v == 512) ? v : 0; }
        if (const char* e = std::getenv("FLS_IVOX_NN_IDS")) nn_ids_mode = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_BALANCED")) balanced = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_HOST_TIMING")) host_timing = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_DEVICE_UPDATE")) allow_device_map = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_DEVICE_MARGIN")) { const long c = std::atol(e); if (c >= 0) device_margin = size_t(c); }
        if (const char* e = std::getenv("FLS_IVOX_DEVICE_EVICT")) device_evict = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_FUSED_UPDATE")) fused_update = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_SHORT_CHAIN")) short_chain_update = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_HOST_SRC")) host_src = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_SPECULATIVE")) speculative_update = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_FUSED_COMMIT")) fused_commit = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_IVOX_FINISH_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) finish_blocks = v; }
        d_upd_state.reserve(1);
        FLS_HIP(hipHostMalloc((void**)&upd_mb_host, sizeof(IvoxUpdMailbox), hipHostMallocMapped));
        std::memset(upd_mb_host, 0, sizeof(IvoxUpdMailbox));
        FLS_HIP(hipHostGetDevicePointer((void**)&upd_mb_dev, upd_mb_host, 0));
        if (const char* e = std::getenv("FLS_IVOX_XCD_CHUNK")) { const int c = std::atoi(e); if (c >= 1 && c <= 4096) xcd_chunk = c; }
        d_ticket.reserve(kTicketWords);
        FLS_HIP(hipMemsetAsync(d_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));
        if (const char* e = std::getenv("FLS_TICKET_SHARDS")) ticket_shards = std::atoi(e) > 1 ? 8 : 1;
        ivox.resolution = 0.5f;       // InitIVox :53-58
        ivox.inv_resolution = 1.0f / 0.5f;
        ivox.capacity = 1000000;
        if (const char* e = std::getenv("FLS_IVOX_CAPACITY")) { const long c = std::atol(e); if (c > 1) ivox.capacity = size_t(c); }  // test hook (LRU eviction)
        return FLS_OK;
    }

    // bring the device image up to date with `ivox`: scatter the journal when possible, else re-flatten
    void refresh_image() {
        if (!image_dirty) return;
        image.want_hash = !use_dense;
        if (rebuild_after_replay) { image_built = false; rebuild_after_replay = false; }
        if (image_built && image.collect_incremental(ivox)) {
            if (image.dir_dirty) image.upload_directory(stream);  // the host path created bricks (their slabs are still zero)
            image.scatter_cell_records(stream, upd_stage);
            ++n_incremental;
        } else {
            image.build_from_ivox(ivox, stream, upd_stage);
            if (image.budget_exceeded) use_dense = false;  // brick pool over its byte budget: per-voxel hash table from now on (map_size(132))
            image_built = true;
            ++n_full_rebuilds;
        }
        image_dirty = false;
        enter_device_mode();
    }

    // Hand the map over to the device-side AddPoints: the brick image has no extent limit, so the only conditions left are the A/B
    // switches (and, without device evictions, a margin below the LRU capacity).
    void enter_device_mode() {
        device_map = false;
        if (!allow_device_map || borrowed || !use_dense || !image.have_bricks || image.want_hash || p.is_localization_mode) return;
        if (!device_evict && ivox.n_alive + device_margin >= ivox.capacity) return;  // without device evictions: stay clear of the capacity
        if (ivox.capacity < 4) return;
        const unsigned long long stamp_base = image.upload_update_meta(ivox, stream, upd_stage);
        IvoxUpdState st{};
        st.n_points = ivox.n_points; st.used = image.used; st.garbage = image.garbage; st.stamp_base = stamp_base;
        st.pts_capacity = image.d_pts.cap; st.n_alive = unsigned(ivox.n_alive); st.lru_capacity = unsigned(std::min<size_t>(ivox.capacity, 0xffffffffu));
        st.next_id = ivox.next_id;
        st.n_bricks = unsigned(image.n_bricks());
        FLS_HIP(hipMemcpyAsync(d_upd_state.p, &st, sizeof(st), hipMemcpyHostToDevice, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        dev_n_points = ivox.n_points; dev_n_alive = ivox.n_alive; dev_n_bricks = image.n_bricks();
        stamp_bound = stamp_base;
        device_map = true;
    }

    // The device image back into the host mirror (device mode ends): the alive voxels as records (one compaction kernel), their
    // points, the LRU order from the stamps, and the bricks the device created.
    void sync_host_from_device() {
        if (!device_map) return;
        IvoxUpdState st{};
        FLS_HIP(hipMemcpyAsync(&st, d_upd_state.p, sizeof(st), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        const size_t nb = std::min<size_t>(st.n_bricks, image.n_bricks_cap), ncell = nb * kBrickStride, n_alive = st.n_alive;
        image.d_alive_rec.reserve(std::max<size_t>(n_alive, 1));
        image.d_counter.reserve(1);
        FLS_HIP(hipMemsetAsync(image.d_counter.p, 0, sizeof(unsigned), stream));
        if (ncell)
            hipLaunchKernelGGL(ivox_list_alive_kernel, dim3(unsigned((ncell + 255) / 256)), dim3(256), 0, stream, (const uint2*)image.d_cells.p,
                               (const unsigned long long*)image.d_brick_key.p, unsigned(ncell), (const unsigned char*)image.d_cap_log2.p,
                               (const unsigned long long*)image.d_stamp.p, image.d_alive_rec.p, image.d_counter.p, unsigned(n_alive));
        FLS_HIP(hipGetLastError());
        unsigned n_listed = 0;
        std::vector<IvoxAliveRec> recs(n_alive);
        std::vector<Pt4> pts(st.used);
        FLS_HIP(hipMemcpyAsync(&n_listed, image.d_counter.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        if (n_alive) FLS_HIP(hipMemcpyAsync(recs.data(), image.d_alive_rec.p, n_alive * sizeof(IvoxAliveRec), hipMemcpyDeviceToHost, stream));
        if (st.used) FLS_HIP(hipMemcpyAsync(pts.data(), image.d_pts.p, st.used * sizeof(Pt4), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        if (n_listed != n_alive) throw std::runtime_error("iVox image: the alive-voxel count of the device state does not match its cells");
        image.download_directory(nb, stream);
        std::vector<HostIvox::ImageVoxel> vox;
        vox.reserve(n_alive);
        for (const IvoxAliveRec& r : recs) vox.push_back(HostIvox::ImageVoxel{r.key, r.begin, r.count, r.cap_log2 ? (1u << r.cap_log2) : 0u, r.stamp});
        const float res = ivox.resolution, inv = ivox.inv_resolution;
        const size_t capacity = ivox.capacity;
        ivox.rebuild_from_image(vox, pts.data(), size_t(st.n_points), st.next_id);
        ivox.resolution = res; ivox.inv_resolution = inv; ivox.capacity = capacity;
        image.used = size_t(st.used);
        image.garbage = size_t(st.garbage);
        image.n_pts_live = size_t(st.n_points);
        device_map = false;
    }

    // One batch of the resident scan through the device-side AddPoints.  Returns true when the device applied it.
    bool device_add_points(const size_t n, const bool counted_by_decide) {
        if (!enqueue_update_chain(n, counted_by_decide)) return false;
        return await_update_chain(n);
    }
    // the launches of one device AddPoints batch (no waiting); false: the batch is too large for the device path
    bool enqueue_update_chain(const size_t n, const bool counted_by_decide) {
        const int nb = int((n + kUpdBlock - 1) / kUpdBlock);
        if (nb > kUpdMaxBlocks) return false;
        d_lx.reserve(n); d_bt.reserve(size_t(nb)); d_seq_src.reserve(n); d_seq_cell.reserve(n); d_jj.reserve(n); d_tlist.reserve(n);
        d_px.reserve(n); d_bt2.reserve(size_t(nb)); d_fbit.reserve(n);
        const IvoxUpdBatch b{d_code.p, d_pw.p, int(n), d_lx.p, d_bt.p, d_seq_src.p, d_seq_cell.p, d_jj.p, d_px.p, d_bt2.p, d_fbit.p, d_tlist.p};
        const IvoxUpdArrays a{image.d_cells.p, image.d_pts.p, image.d_cap_log2.p, image.d_stamp.p, image.d_pend.p, image.d_rank_mm.p,
                              image.d_dir.p, image.dir_mask, unsigned(image.n_bricks_cap), image.d_brick_key.p, image.d_nbr.p, ivox.inv_resolution};
        upd_seq = (upd_seq + 1u) & 0x7fffffffu;
        if (upd_seq == 0u) upd_seq = 1u;
        const dim3 g{unsigned(nb), 1u, 1u}, t{unsigned(kUpdBlock), 1u, 1u};
        // three forms of the same phases (kernels_ivox_update.hpp): the SHORT chain (default: five launches behind the decision), the
        // one-workgroup single launch (FLS_IVOX_FUSED_UPDATE=1, small batches; measured slower: one CU's latency chains), and the LONG
        // chain, which batches that may reach the LRU capacity need (the eviction selection has grid-wide steps of its own)
        const bool cannot_evict = dev_n_alive + n < ivox.capacity;
        const bool fused = fused_update && n <= size_t(kFusedMaxN) && cannot_evict;
        const bool short_chain = !fused && short_chain_update && cannot_evict && counted_by_decide;
        if (fused) {
            hipLaunchKernelGGL(ivox_upd_fused_kernel, dim3(1), dim3(kFusedThreads), 0, stream, b, a, d_upd_state.p, upd_mb_dev, upd_seq);
            ++n_fused_updates;
        } else if (short_chain) {
            hipLaunchKernelGGL(ivox_upd_seq_nb, g, t, 0, stream, b, a, d_upd_state.p, nb);
            hipLaunchKernelGGL(ivox_upd_plan, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
            hipLaunchKernelGGL(ivox_upd_last_regions, g, t, 0, stream, b, a, d_upd_state.p);
            hipLaunchKernelGGL(ivox_upd_points, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
            if (fused_commit) {
                const unsigned fb = unsigned(std::min<size_t>(size_t(finish_blocks), (n + kUpdBlock / 64 - 1) / (kUpdBlock / 64)));
                hipLaunchKernelGGL(ivox_upd_finish_commit, dim3(fb), t, 0, stream, b, a, d_upd_state.p, upd_mb_dev, upd_seq, d_ticket.p);
            } else {
                hipLaunchKernelGGL(ivox_upd_finish, dim3(unsigned((n + kUpdBlock / 64 - 1) / (kUpdBlock / 64))), t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
                hipLaunchKernelGGL(ivox_upd_commit, dim3(1), dim3(64), 0, stream, d_upd_state.p, upd_mb_dev, upd_seq, unsigned(image.n_bricks_cap));
            }
            ++n_short_updates;
        } else {
        hipLaunchKernelGGL(ivox_upd_count, g, t, 0, stream, b);
        hipLaunchKernelGGL(ivox_upd_scan1, dim3(1), dim3(kUpdMaxBlocks), 0, stream, b, nb, d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_seq, g, t, 0, stream, b, a, d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_plan, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
        // LRU evictions inside the batch: whenever the batch COULD reach the capacity (every point a new voxel), the alive cells are
        // listed and sorted by their 64-bit stamp (two stable 32-bit radix rounds) so that scan2 / ivox_evict_check can pick the tail
        const bool may_evict = device_evict && dev_n_alive + n >= ivox.capacity && dev_n_alive > 0;
        unsigned n_list = 0;
        if (may_evict) {
            const unsigned ncell = unsigned(dev_n_bricks * kBrickStride), nbe = (ncell + kEvBlock - 1) / kEvBlock;  // (bricks this batch creates hold no candidate)
            n_list = unsigned(dev_n_alive);
            d_ev_bt.reserve(size_t(2) * nbe);
            ev_sort.prepare(n_list);
            hipLaunchKernelGGL(ivox_evict_count, dim3(nbe), dim3(kEvBlock), 0, stream, (const uint2*)image.d_cells.p, ncell, d_ev_bt.p);
            hipLaunchKernelGGL(vg_scan, dim3(1), dim3(kVgScanBlock), 0, stream, (const unsigned*)d_ev_bt.p, d_ev_bt.p + nbe, int(nbe), (unsigned*)nullptr);
            hipLaunchKernelGGL(ivox_evict_list, dim3(nbe), dim3(kEvBlock), 0, stream, (const uint2*)image.d_cells.p, (const unsigned long long*)image.d_stamp.p, ncell,
                               (const unsigned*)(d_ev_bt.p + nbe), ev_sort.k0, ev_sort.v0);
            ev_sort.run(4, stream);
            hipLaunchKernelGGL(ivox_evict_hikeys, dim3((n_list + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long*)image.d_stamp.p,
                               (const unsigned*)ev_sort.v0, n_list, ev_sort.k0);
            ev_sort.run(DevicePairSort::passes_for((unsigned long long)((stamp_bound + n) >> 32) + 1ull), stream);
        }
        hipLaunchKernelGGL(ivox_upd_scan2, dim3(1), dim3(kUpdMaxBlocks), 0, stream, b, d_upd_state.p, may_evict ? 1u : 0u, n_list);  // (also decides when no eviction selection follows)
        if (may_evict) {
            d_crank.reserve(n);
            d_evict_list.reserve(n);
            hipLaunchKernelGGL(ivox_upd_cranks, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p, d_crank.p);
            hipLaunchKernelGGL(ivox_evict_select, dim3(1), dim3(kEvBlock), 0, stream, (const unsigned*)ev_sort.v0, a, d_upd_state.p, (const unsigned*)d_crank.p, d_evict_list.p);
            // voxels the selection evicts BEFORE their first point of this batch arrives are re-created by it (round 4; such a batch used to
            // be refused): the plan runs again seeing them as creations, and the totals / block offsets / point-array check with it
            hipLaunchKernelGGL(ivox_upd_plan, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
            hipLaunchKernelGGL(ivox_upd_scan2_again, dim3(1), dim3(kUpdMaxBlocks), 0, stream, b, d_upd_state.p);
        }
        if (may_evict) hipLaunchKernelGGL(ivox_upd_decide, dim3(1), dim3(64), 0, stream, d_upd_state.p);  // (the selection may still refuse the batch)
        if (may_evict) hipLaunchKernelGGL(ivox_evict_apply, g, t, 0, stream, (const unsigned*)d_evict_list.p, a, d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_last, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_regions, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_points, g, t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_finish, dim3(unsigned((n + kUpdBlock / 64 - 1) / (kUpdBlock / 64))), t, 0, stream, b, a, (const IvoxUpdState*)d_upd_state.p);
        hipLaunchKernelGGL(ivox_upd_commit, dim3(1), dim3(64), 0, stream, d_upd_state.p, upd_mb_dev, upd_seq, unsigned(image.n_bricks_cap));
        }
        FLS_HIP(hipGetLastError());
        return true;
    }
    // the verdict of the batch queued last; true: applied.  (kUpdSkipped -- a speculative chain that found nothing to do -- reads as "not applied".)
    bool await_update_chain(const size_t n) {
        // (a few words in host-mapped memory; no copy, no stream synchronisation)
        for (unsigned long long spin = 1;; ++spin) {
            if (__atomic_load_n(&upd_mb_host->seq, __ATOMIC_ACQUIRE) == upd_seq) break;
            if ((spin & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) FLS_HIP(q);
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        dev_n_bricks = std::min<size_t>(upd_mb_host->n_bricks, image.n_bricks_cap);  // (bricks are created whatever the verdict)
        last_chain_skipped = (upd_mb_host->status & kUpdSkipped) != 0u;
        if (last_chain_skipped) return false;
        if (upd_mb_host->status != kUpdOk) {
            if (upd_mb_host->status & kUpdEvictConflict) ++n_refused_conflict;
            if (upd_mb_host->status & kUpdArrayFull) { ++n_refused_full; rebuild_after_replay = true; }  // point array or brick pool: re-flatten with more room
            if (upd_mb_host->status & kUpdOutside) ++n_refused_outside;
            return false;
        }
        dev_n_points = size_t(upd_mb_host->n_points);
        dev_n_alive = size_t(upd_mb_host->n_alive);
        stamp_bound += n;
        n_device_evictions += upd_mb_host->evicted;
        n_device_recreated += upd_mb_host->recreated;
        ++n_device_updates;
        return true;
    }

    // pcl::transformPoint with Affine3d(T_): double evaluation, float result
    static PtI xform_d(const PtI& p, const double* T) {
        PtI r = p;
        const double x = p.x, y = p.y, z = p.z;
        r.x = float(((T[0] * x + T[4] * y) + T[8] * z) + T[12]);
        r.y = float(((T[1] * x + T[5] * y) + T[9] * z) + T[13]);
        r.z = float(((T[2] * x + T[6] * y) + T[10] * z) + T[14]);
        return r;
    }

    // ids form -> rows form for every point (the ids are slots of the image as it is NOW: call before anything moves slots)
    void ensure_nn_rows() {
        if (nn_rows_current || nn_n == 0) { nn_rows_current = true; return; }
        const IvoxImage& im = borrowed ? *borrowed : image;
        hipLaunchKernelGGL(ivox_nn_materialize_kernel, dim3(unsigned((nn_n + 255) / 256)), dim3(256), 0, stream, (const unsigned*)d_nn_ids.p, d_nn_cnt.p, int(nn_n),
                           (const float4*)im.d_pts.p, unsigned(im.d_pts.cap), d_nn.p);  // (slot bound = the allocation: in device mode the host's `used` is stale)
        FLS_HIP(hipGetLastError());
        nn_rows_current = true;
    }
    void download_nn() {
        ensure_nn_rows();
        const size_t n = nn_n;
        h_nn.resize(n * 5);
        h_cnt.resize(n);
        if (n == 0) return;
        FLS_HIP(hipMemcpyAsync(h_nn.data(), d_nn.p, n * 5 * sizeof(float4), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipMemcpyAsync(h_cnt.data(), d_nn_cnt.p, n, hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        for (unsigned char& c : h_cnt) c &= 7;  // (bit 7 = rows form)
    }

    fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) override {
        if (c1 != nullptr && n1 != 0) return FLS_ERR_INVALID;  // CHECK_EQ(cloud_list.size(), 1) :61
        const std::vector<PtI> cloud = cloud_from(c0, n0, stride);
        return add_cloud_impl(cloud);
    }

    // the decision launch (ivox_add_decide_kernel): codes + world points of the resident scan; in device mode it also counts the codes and
    // opens the batch.  speculative: gated on the device by the Gauss-Newton state, pose read from it.  Returns "counted".
    bool launch_decide(const size_t n, const bool speculative) {
        d_code.reserve(n);
        d_pw.reserve(n);
        Pose16 Tw;
        std::memcpy(Tw.m, T_, sizeof(Tw.m));
        const GnState* const gn = speculative ? (const GnState*)d_state.p : nullptr;
        const int max_it = int(p.max_iterations);
        // device mode: the decision launch also counts the insertion codes per block and opens the batch (ivox_upd_count's job)
        const int nb = int((n + kUpdBlock - 1) / kUpdBlock);
        const bool count_here = device_map && nb <= kUpdMaxBlocks;
        if (count_here) { d_lx.reserve(n); d_bt.reserve(size_t(nb)); }
        uint2* const lx_arg = count_here ? d_lx.p : nullptr;
        uint2* const bt_arg = count_here ? d_bt.p : nullptr;
        unsigned* const st_status = count_here ? &d_upd_state.p->status : nullptr;
        unsigned* const st_apply = count_here ? &d_upd_state.p->apply : nullptr;
        // (the update moves map slots: lists still in ids form become rows in the same launch)
        if (nn_ids_mode && (speculative || !nn_rows_current) && nn_n > 0) {
            const size_t m = std::max(n, nn_n);
            hipLaunchKernelGGL(ivox_add_decide_kernel<true>, dim3(unsigned((m + 255) / 256)), dim3(256), 0, stream, scan.x.p, scan.y.p, scan.z.p,
                               int(n), Tw, d_nn.p, d_nn_cnt.p, int(nn_n), filter_size_map_min,
                               d_code.p, d_pw.p, (const unsigned*)d_nn_ids.p, (const float4*)image.d_pts.p, unsigned(image.d_pts.cap), lx_arg, bt_arg, st_status, st_apply, gn, max_it);
            if (!speculative) nn_rows_current = true;
        } else {
            hipLaunchKernelGGL(ivox_add_decide_kernel<false>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, scan.x.p, scan.y.p, scan.z.p,
                               int(n), Tw, d_nn.p, d_nn_cnt.p, int(nn_n), filter_size_map_min,
                               d_code.p, d_pw.p, (const unsigned*)(nn_ids_mode ? d_nn_ids.p : nullptr), (const float4*)image.d_pts.p, unsigned(image.d_pts.cap), lx_arg, bt_arg,
                               st_status, st_apply, gn, max_it);
        }
        FLS_HIP(hipGetLastError());
        return count_here;
    }

    fls_status add_cloud_impl(const std::vector<PtI>& planar_cloud, const bool from_resident_scan = false, const bool pre_enqueued = false) {
        if (replica_only) return FLS_ERR_STATE;
        if (p.is_localization_mode) { device_map = false; is_first = true; ivox.clear(); image_built = false; }
        fls_status rc = FLS_OK;
        if (!(from_resident_scan && !is_first)) { ensure_nn_rows(); sync_host_from_device(); }  // every other branch works on the host mirror
        if (is_first) {
            rc = ivox.add_points(planar_cloud.data(), planar_cloud.size());
            if (rc != FLS_OK) return rc;
            is_first = false;
        } else if (from_resident_scan) {
            // :79-131 on the device (the cloud is the scan just matched, still resident): decision code + world point
            // per source point; the host only walks the codes to build the two insertion lists in index order.
            std::vector<PtI> to_add, no_downsample;
            const size_t n = std::min(number_planar_point, scan.n);
            const auto tm0 = std::chrono::steady_clock::now();
            bool counted = false;
            if (n) {
                if (!pre_enqueued) {
                    counted = launch_decide(n, /*speculative=*/false);
                    ensure_nn_rows();
                    if (device_map && !enqueue_update_chain(n, counted)) { ++n_host_fallbacks; sync_host_from_device(); }
                } else {
                    counted = true;
                }
                if (device_map) {
                    if (await_update_chain(n)) {
                        if (host_timing)
                            std::fprintf(stderr, "[fls host] device AddPoints: %u points into %u voxels, %.3f ms\n", upd_mb_host->added, upd_mb_host->touched,
                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count());
                        return FLS_OK;
                    }
                    if (last_chain_skipped) {
                        // a speculative chain that judged "no update here" although the host wants one (cannot happen while host and device
                        // read the same words; kept as a safe path): decide + apply the ordinary way
                        if (nn_ids_mode) nn_rows_current = false;  // (the skipped launch materialised nothing)
                        counted = launch_decide(n, false);
                        ensure_nn_rows();
                        if (enqueue_update_chain(n, counted) && await_update_chain(n)) return FLS_OK;
                    }
                    // refused (an eviction-order conflict, the point array or brick pool full): nothing was applied; the exact sequential
                    // path below replays the batch on the host mirror
                    ++n_host_fallbacks;
                    sync_host_from_device();
                }
                h_code.resize(n);  // (host copies only on this path: the device applied nothing)
                h_pw.resize(n);
                FLS_HIP(hipMemcpyAsync(h_code.data(), d_code.p, n, hipMemcpyDeviceToHost, stream));
                FLS_HIP(hipMemcpyAsync(h_pw.data(), d_pw.p, n * sizeof(float4), hipMemcpyDeviceToHost, stream));
                FLS_HIP(hipStreamSynchronize(stream));
                for (size_t i = 0; i < n; ++i) {
                    if (h_code[i] == 0) continue;
                    const PtI pw{h_pw[i].x, h_pw[i].y, h_pw[i].z, scan.staged_intensity(i)};
                    (h_code[i] == 1 ? to_add : no_downsample).push_back(pw);
                }
            }
            const auto tm1 = std::chrono::steady_clock::now();
            rc = ivox.add_points(to_add.data(), to_add.size());
            if (rc != FLS_OK) return rc;
            rc = ivox.add_points(no_downsample.data(), no_downsample.size());
            if (rc != FLS_OK) return rc;
            if (host_timing) {
                const auto tm2 = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[fls host] decide+lists %.3f ms (%zu + %zu pts), AddPoints %.3f ms\n",
                             std::chrono::duration<double, std::milli>(tm1 - tm0).count(), to_add.size(), no_downsample.size(),
                             std::chrono::duration<double, std::milli>(tm2 - tm1).count());
            }
        } else {
            // external non-first call with an arbitrary cloud: same rule on the host (:79-131)
            download_nn();
            std::vector<PtI> to_add, no_downsample;
            const double fs = filter_size_map_min, half = 0.5 * filter_size_map_min;
            const size_t n = std::min(number_planar_point, planar_cloud.size());
            for (size_t i = 0; i < n; ++i) {
                const PtI pw = xform_d(planar_cloud[i], T_);
                const int cnt = i < nn_n ? h_cnt[i] : 0;
                if (cnt > 0) {
                    const Pt4* near = &h_nn[i * 5];
                    const double c[3] = {(std::floor(double(pw.x) / fs) + 0.5) * fs, (std::floor(double(pw.y) / fs) + 0.5) * fs,
                                         (std::floor(double(pw.z) / fs) + 0.5) * fs};
                    const double d0[3] = {double(near[0].x) - c[0], double(near[0].y) - c[1], double(near[0].z) - c[2]};
                    if (std::fabs(d0[0]) > half && std::fabs(d0[1]) > half && std::fabs(d0[2]) > half) {
                        no_downsample.push_back(pw);
                        continue;
                    }
                    bool need_add = true;
                    const double e[3] = {double(pw.x) - c[0], double(pw.y) - c[1], double(pw.z) - c[2]};
                    const double dist = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
                    if (cnt >= 5) {
                        for (int k = 0; k < 5; ++k) {
                            const double f[3] = {double(near[k].x) - c[0], double(near[k].y) - c[1], double(near[k].z) - c[2]};
                            if ((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2] < dist + 1.0e-6) { need_add = false; break; }
                        }
                    }
                    if (need_add) to_add.push_back(pw);
                } else {
                    to_add.push_back(pw);
                }
            }
            rc = ivox.add_points(to_add.data(), to_add.size());
            if (rc != FLS_OK) return rc;
            rc = ivox.add_points(no_downsample.data(), no_downsample.size());
            if (rc != FLS_OK) return rc;
        }
        image_dirty = true;
        {
            const auto tr0 = std::chrono::steady_clock::now();
            refresh_image();
            if (host_timing)
                std::fprintf(stderr, "[fls host] refresh_image %.3f ms (%zu pt updates, %zu cell updates)\n",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), image.pt_upd.size(),
                             image.cell_upd.size());
        }
        if (p.is_localization_mode) {  // :134-138 kd-tree for GetFitnessScore
            rc = fitness_grid.build(planar_cloud, 1.0f, stream);
            have_fitness_grid = (rc == FLS_OK);
        }
        return rc;
    }

    // fls_match from host buffers: the scan stays in the pinned staging buffer; the first iteration's correspondence launch reads it
    // from there over PCIe (the kernel is latency bound: the reads hide behind its probe / candidate chains) and leaves the device copy
    // the later launches use.  No SDMA copy, no copy-engine -> compute-queue hand-off in front of the first kernel (8 us + 8 us for the
    // 10 k-point planar cloud the pipeline feeds; FLS_IVOX_HOST_SRC=0 restores the copy).
    bool host_src = true, scan_in_staging = false;
    fls_status scan_upload_for_match(const float* s0, size_t n0, const float* s1, size_t n1, int stride) override {
        if (!host_src || borrowed) return scan_upload(s0, n0, s1, n1, stride);
        (void)s1; (void)n1;
        scan.stage_raw(s0, n0, stride);
        scan_in_staging = n0 != 0;
        if (scan_in_staging) scan.reserve_device();
        return FLS_OK;
    }
    fls_status scan_upload(const float* s0, size_t n0, const float* s1, size_t n1, int stride) override {
        scan_in_staging = false;
        (void)s1; (void)n1;
        // (the pinned staging buffer is free again: fls_scan_upload synchronises, a Match ends after its copies)
        scan.upload_raw(s0, n0, stride, stream);
        return FLS_OK;
    }

    // e0 / e1 (profiling only): start / stop events attached to the kernel's own dispatch packet (hipExtLaunchKernelGGL),
    // i.e. the kernel's execution time as a kernel trace sees it -- a hipEventRecord bracket also times the dispatch of
    // the kernel between its two marker packets (about 3 us more on an 18 us kernel)
    template <int G>
    void launch_knn(const size_t n, const int first, const Pose16& T0, const DevGrid& g, const BrickDir& win, hipEvent_t e0 = nullptr,
                    hipEvent_t e1 = nullptr) {
        unsigned* const ids_arg = nn_ids_mode ? d_nn_ids.p : nullptr;
        // first launch of a Match whose scan is still in the staging buffer: read it there, write the device copy
        const bool from_host = first && scan_in_staging;
        const float* const hx = from_host ? scan.stage_dev() : nullptr;
        const float* const src_x = from_host ? hx : (const float*)scan.x.p;
        const float* const src_y = from_host ? hx + n : (const float*)scan.y.p;
        const float* const src_z = from_host ? hx + 2 * n : (const float*)scan.z.p;
        float* const dev_copy = from_host ? scan.xyz.p : nullptr;
        const size_t nblk = (n * G + 255) / 256, gran = size_t(8) * size_t(xcd_chunk);
        const dim3 grid(unsigned((nblk + gran - 1) / gran * gran));  // multiple of 8 * chunk: the XCD re-map is a bijection
#define FLS_KNN_L(C, D, F, B)                                                                                                        \
    do {                                                                                                                             \
        if (e0 || !plain_launch)                                                                                                     \
            hipExtLaunchKernelGGL((ivox_knn_kernel<G, C, D, F, B>), grid, dim3(256), 0, stream, e0, e1, 0, src_x,                    \
                                  src_y, src_z, int(n), (const GnState*)d_state.p, T0, g, win,                                      \
                                  ivox.inv_resolution, d_nn.p, d_nn_cnt.p, d_flag.p, d_tc.p, xcd_chunk, ids_arg, nn_prev, dev_copy); \
        else                                                                                                                         \
            hipLaunchKernelGGL((ivox_knn_kernel<G, C, D, F, B>), grid, dim3(256), 0, stream, src_x,                                  \
                               src_y, src_z, int(n), (const GnState*)d_state.p, T0, g, win,                                         \
                               ivox.inv_resolution, d_nn.p, d_nn_cnt.p, d_flag.p, d_tc.p, xcd_chunk, ids_arg, nn_prev, dev_copy);   \
    } while (0)
#define FLS_KNN(C, D)                                                                                                                \
    do {                                                                                                                             \
        if (balanced && G == 4) { if (first) FLS_KNN_L(C, D, true, true); else FLS_KNN_L(C, D, false, true); }                        \
        else { if (first) FLS_KNN_L(C, D, true, false); else FLS_KNN_L(C, D, false, false); }                                         \
    } while (0)
        if (win.cells) { if (count_traffic) FLS_KNN(true, true); else FLS_KNN(false, true); }
        else { if (count_traffic) FLS_KNN(true, false); else FLS_KNN(false, false); }
#undef FLS_KNN_L
#undef FLS_KNN
    }

    fls_status match_resident(double* T, int update_map, fls_stats* out) override {
        if (replica_only && update_map) return FLS_ERR_STATE;  // (a replica's image has no AddPoints side and no mirror)
        const size_t n = scan.n;
        number_planar_point = n;
        stats = fls_stats{};
        stats.n_source = int(n);
        std::memcpy(T_, T, sizeof(T_));
        if (n == 0) {
            // empty planar cloud: H = g = 0 -> dx = 0 -> stop rule fires in iteration 0, n_valid = 0 < 50 -> false (:201-203)
            stats.iterations = 1; stats.converged = 0;
            if (out) *out = stats;
            // the iteration log of that one iteration (fls_get_iteration_log; the oracle keeps the same row): the input pose, no valid point, no residual
            log_n = 0; log_stale = false;
            if (h_state.p) {
                std::memcpy(h_state.p->log_T[0], T, sizeof(double) * 16);
                h_state.p->log_nv[0] = 0; h_state.p->log_res[0] = 0.0; h_state.p->iter = 1;
                log_n = 1;
            }
            return FLS_NOT_CONVERGED;
        }
        if (!borrowed) refresh_image();
        const IvoxImage& im = borrowed ? *borrowed : image;
        // nearest_points_.resize(n) semantics (:257): grown tail is empty, shrink forgets
        d_nn.reserve(n * 5, /*keep=*/true, stream);
        d_nn_cnt.reserve(n, /*keep=*/true, stream);
        if (nn_ids_mode) d_nn_ids.reserve(n * 8, /*keep=*/true, stream);
        nn_prev = int(std::min(nn_n, n));  // the first iteration's kNN launch clears the counts of the grown tail [nn_prev, n)
        nn_n = n;
        d_J.reserve(7 * n);
        d_flag.reserve(n);  // cleared by the first iteration's kNN kernel (std::fill(flags, false) once per Match, :156, Q1)
        // workgroup size of the fit kernel: 512 threads (two waves per SIMD of a CU) when the scan fills the machine anyway; 256 (one wave per
        // SIMD) up to 65,536 points, where 256 workgroups spread the same waves over every CU -- the 9.8k-point planar cloud the pipeline feeds
        // ran its 2,500-instruction fit phase two waves deep on 20 CUs with 236 CUs idle
        const int fit_threads = (fit_threads_env > 0) ? fit_threads_env : (n <= 65536 ? 256 : kFitThreads);
        const int nwg = int((n + size_t(fit_threads) - 1) / size_t(fit_threads));
        d_partials_b.reserve(size_t(nwg) * kPartialStride);
        const int iters = int(p.max_iterations);
        const DevGrid g = im.dev();
        const BrickDir win = use_dense ? im.bricks() : BrickDir{nullptr, 0u, nullptr, 0u};
        Pose16 T0;
        std::memcpy(T0.m, T, sizeof(T0.m));
        // speculative map update (see speculative_update): only where the short chain applies and the call would update the map if it converges
        const bool spec = speculative_update && update_map && !p.is_localization_mode && !borrowed && !is_first && device_map && short_chain_update && !fused_update &&
                          dev_n_alive + n < ivox.capacity && int((n + kUpdBlock - 1) / kUpdBlock) <= kUpdMaxBlocks;
        spec_pending = false;
        auto after_chunk = [&](int) {
            if (!spec) return;
            if (spec_pending) ++n_speculative_skipped;  // (the chain behind the previous chunk found the Match unfinished)
            const bool counted = launch_decide(n, /*speculative=*/true);
            spec_pending = counted && enqueue_update_chain(n, counted);
            ++n_speculative;
        };
        const unsigned word = run_mailbox_loop(iters, n, [&](int it, int first) {
            hipEvent_t e0 = profiling ? ev[2 * it] : nullptr, e1 = profiling ? ev[2 * it + 1] : nullptr;
            hipEvent_t f0 = nullptr, f1 = nullptr;
            if (prof_fit) { f0 = e0; f1 = e1; e0 = e1 = nullptr; }
            if (variant == 4) launch_knn<4>(n, first, T0, g, win, e0, e1); else launch_knn<8>(n, first, T0, g, win, e0, e1);
#define FLS_FIT_NT(F, NT)                                                                                                            \
    do {                                                                                                                             \
        if (f0 || !plain_launch)                                                                                                     \
            hipExtLaunchKernelGGL((p2plane_fit_solve_kernel<F, NT>), dim3(nwg), dim3(NT), 0, stream, f0, f1, 0, scan.x.p, scan.y.p, scan.z.p, int(n),  \
                       d_state.p, T0, (const float4*)d_nn.p, (const unsigned char*)d_nn_cnt.p, d_J.p, d_flag.p, d_partials_b.p,     \
                       d_ticket.p, mb_dev, launch_word(), p.point_to_planar_thres, p.rotation_converge_thres, p.position_converge_thres, ticket_shards,  \
                       (const unsigned*)(nn_ids_mode ? d_nn_ids.p : nullptr), (const float4*)g.pts, unsigned(im.d_pts.cap));  \
        else                                                                                                                         \
            hipLaunchKernelGGL((p2plane_fit_solve_kernel<F, NT>), dim3(nwg), dim3(NT), 0, stream, scan.x.p, scan.y.p, scan.z.p, int(n),  \
                       d_state.p, T0, (const float4*)d_nn.p, (const unsigned char*)d_nn_cnt.p, d_J.p, d_flag.p, d_partials_b.p,     \
                       d_ticket.p, mb_dev, launch_word(), p.point_to_planar_thres, p.rotation_converge_thres, p.position_converge_thres, ticket_shards,  \
                       (const unsigned*)(nn_ids_mode ? d_nn_ids.p : nullptr), (const float4*)g.pts, unsigned(im.d_pts.cap));  \
    } while (0)
#define FLS_FIT(F) do { if (fit_threads == 256) FLS_FIT_NT(F, 256); else FLS_FIT_NT(F, kFitThreads); } while (0)
            if (first) FLS_FIT(true); else FLS_FIT(false);
#undef FLS_FIT_NT
#undef FLS_FIT
        }, after_chunk);
        scan_in_staging = false;  // (the first launch left the device copy)
        if (nn_ids_mode) nn_rows_current = false;  // the lists of every point with candidates are slots of the current image now
        const Mailbox& mb = *mb_host;
        const int used = int(word & 0xffu);
        std::memcpy(T, mb.T, sizeof(double) * 16);
        std::memcpy(T_, mb.T, sizeof(T_));
        std::memcpy(final_T, mb.T, sizeof(final_T));
        have_final = true;
        bool has_converge = true;
        if (mb.n_valid < 50) has_converge = false;  // :201-203
        stats.iterations = used;
        stats.n_valid = mb.n_valid;
        stats.sum_res = mb.sum_res;
        std::memcpy(stats.last_dx, mb.last_dx, sizeof(stats.last_dx));
        stats.converged = has_converge ? 1 : 0;
        fls_status rc = has_converge ? FLS_OK : FLS_NOT_CONVERGED;
        if (has_converge && !p.is_localization_mode && update_map && !borrowed) {  // :205-206
            if (spec_pending && nn_ids_mode) nn_rows_current = true;  // (the speculative decision launch turned the lists into rows)
            const fls_status arc = add_cloud_impl(scan.host, /*from_resident_scan=*/true, /*pre_enqueued=*/spec_pending);
            if (arc != FLS_OK) rc = arc; else stats.map_updated = 1;
        } else if (spec_pending) {
            ++n_speculative_skipped;  // not converged: the chain skipped itself on the device (n_valid < 50), nothing to collect
        }
        spec_pending = false;
        if (out) *out = stats;
        return rc;
    }

    void reset_job_state() override { nn_n = 0; have_final = false; }  // nearest_points_ of a fresh matcher is empty

    // ---- map image export / import (fls_reg.h): header, voxels from the LRU tail (oldest) to the head {key, count}, then the
    // points {x, y, z, id} of the voxels in the same order
    struct BlobHeader {
        char magic[8];
        unsigned version, kind;
        float resolution;
        unsigned is_first;
        unsigned long long capacity, n_voxels, n_points;
        long long next_id;
    };
    struct BlobVoxel { unsigned long long key; unsigned count, pad; };
    size_t map_export(void* blob, size_t cap) override {
        if (borrowed || replica_only) return 0;
        const bool was_device = device_map;
        sync_host_from_device();
        const size_t need = sizeof(BlobHeader) + ivox.n_alive * sizeof(BlobVoxel) + ivox.n_points * sizeof(Pt4);
        if (blob && cap >= need) {
            char* w = static_cast<char*>(blob);
            BlobHeader hd{};
            std::memcpy(hd.magic, "FLSIVOX1", 8);
            hd.version = 1; hd.kind = unsigned(kind); hd.resolution = ivox.resolution; hd.is_first = is_first ? 1u : 0u;
            hd.capacity = ivox.capacity; hd.n_voxels = ivox.n_alive; hd.n_points = ivox.n_points; hd.next_id = ivox.next_id;
            std::memcpy(w, &hd, sizeof(hd));
            BlobVoxel* bv = reinterpret_cast<BlobVoxel*>(w + sizeof(hd));
            Pt4* bp = reinterpret_cast<Pt4*>(w + sizeof(hd) + ivox.n_alive * sizeof(BlobVoxel));
            size_t k = 0, q = 0;
            for (int v = ivox.tail; v >= 0; v = ivox.pool[v].prev) {
                const HostIvox::Voxel& vx = ivox.pool[v];
                bv[k++] = BlobVoxel{vx.key, unsigned(vx.pts.size()), 0u};
                std::memcpy(bp + q, vx.pts.data(), vx.pts.size() * sizeof(Pt4));
                q += vx.pts.size();
            }
        }
        if (was_device) enter_device_mode();
        return need;
    }
    fls_status map_import(const void* blob, size_t n) override {
        if (borrowed || n < sizeof(BlobHeader)) return FLS_ERR_INVALID;
        const char* r = static_cast<const char*>(blob);
        BlobHeader hd;
        std::memcpy(&hd, r, sizeof(hd));
        if (std::memcmp(hd.magic, "FLSIVOX1", 8) != 0 || hd.version != 1 || hd.kind != unsigned(kind)) return FLS_ERR_INVALID;
        // the header is untrusted (it may come off a broadcast): bound both counts by the payload BEFORE multiplying (no u64 wrap),
        // the resolution must be this handle's, every voxel non-empty and unique, the counts must add up in 64 bits
        static_assert(sizeof(BlobVoxel) == 16 && sizeof(Pt4) == 16, "blob records");
        const unsigned long long payload = (unsigned long long)(n - sizeof(BlobHeader)) / 16ull;
        if ((n - sizeof(BlobHeader)) % 16u != 0 || hd.n_voxels > payload || hd.n_points > payload || hd.n_voxels + hd.n_points != payload) return FLS_ERR_INVALID;
        if (!(hd.resolution > 0.f) || !std::isfinite(hd.resolution) || hd.resolution != ivox.resolution) return FLS_ERR_INVALID;
        if (hd.n_voxels > hd.n_points || hd.next_id < 0 || (unsigned long long)hd.next_id < hd.n_points) return FLS_ERR_INVALID;
        const BlobVoxel* bv = reinterpret_cast<const BlobVoxel*>(r + sizeof(hd));
        const Pt4* bp = reinterpret_cast<const Pt4*>(r + sizeof(hd) + hd.n_voxels * sizeof(BlobVoxel));
        std::vector<HostIvox::ImageVoxel> vox(size_t(hd.n_voxels));
        unsigned long long qsum = 0;
        for (size_t k = 0; k < vox.size(); ++k) {  // stamp = position in the LRU order (tail first)
            if (bv[k].count == 0u || qsum + bv[k].count > hd.n_points) return FLS_ERR_INVALID;
            vox[k] = HostIvox::ImageVoxel{bv[k].key, unsigned(qsum), bv[k].count, 0u, (unsigned long long)(k + 1)};
            qsum += bv[k].count;
        }
        if (qsum != hd.n_points) return FLS_ERR_INVALID;
        {
            std::vector<unsigned long long> keys(vox.size());
            for (size_t k = 0; k < vox.size(); ++k) keys[k] = vox[k].key;
            std::sort(keys.begin(), keys.end());
            if (std::adjacent_find(keys.begin(), keys.end()) != keys.end()) return FLS_ERR_INVALID;  // a voxel listed twice
        }
        device_map = false;
        replica_only = false;
        const size_t capacity = ivox.capacity;
        ivox.rebuild_from_image(vox, bp, size_t(hd.n_points), int(hd.next_id));
        ivox.resolution = hd.resolution; ivox.inv_resolution = 1.0f / hd.resolution; ivox.capacity = capacity;
        is_first = hd.is_first != 0;
        nn_n = 0; have_final = false; nn_rows_current = true;
        image_built = false;  // the slot layout is rebuilt from the mirror (window order), like the first build of the exporter
        image_dirty = true;
        refresh_image();
        return FLS_OK;
    }

    // ---- replica sets: the owner's device image copied device to device (hipMemcpyPeer over xGMI between GPUs; no export blob, no host
    // mirror rebuild, no re-flatten -- SURVEY 8e's replicated read-only map).  Both devices' streams are idle when this returns.
    bool can_replicate() const override { return !borrowed; }
    size_t live_counts(size_t& n_bricks_live) {  // slots in use + bricks, from the device's own state while it maintains the map
        if (!device_map) { n_bricks_live = image.n_bricks(); return image.used; }
        IvoxUpdState st{};
        FLS_HIP(hipMemcpyAsync(&st, d_upd_state.p, sizeof(st), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        n_bricks_live = std::min<size_t>(st.n_bricks, image.n_bricks_cap);
        return size_t(st.used);
    }
    fls_status replicate_from(fls_matcher& o) override {
        if (borrowed || o.kind != kind) return FLS_ERR_INVALID;
        auto& src = static_cast<P2PlaneIvoxMatcher&>(o);
        if (src.borrowed || src.replica_only || src.ivox.resolution != ivox.resolution) return FLS_ERR_STATE;
        FLS_HIP(hipSetDevice(src.device));
        const fls_status prc = src.prepare_batch();  // image current, the owner's stream idle
        if (prc != FLS_OK) return prc;
        size_t nb = 0;
        const size_t used_now = src.live_counts(nb);
        FLS_HIP(hipSetDevice(device));
        FLS_HIP(hipStreamSynchronize(stream));
        device_map = false;
        replica_only = false;
        ivox.clear();
        image_built = false; image_dirty = true;  // (a copy that fails half-way leaves an empty map that rebuilds its image)
        use_dense = src.use_dense;
        image.clone_for_reading(src.image, used_now, nb, src.device, device, stream);
        is_first = src.is_first;
        replica_only = true;
        image_built = true; image_dirty = false; rebuild_after_replay = false;
        nn_n = 0; have_final = false; nn_rows_current = true; spec_pending = false;
        return FLS_OK;
    }

    // ---- the device image as one flat buffer, for replicas in OTHER processes (torch.distributed ranks): fls_map_image_bytes / _export / _import.
    // The exporter keeps its map; the importer becomes a read-only replica (fls_match / fls_match_batch with update_map == 0), like a
    // member of a replica set.  No host mirror, no re-flatten: 44 MB for the 1e6-point map, copied at memory speed on either side.
    size_t map_image_bytes() override {
        if (borrowed || replica_only) return 0;
        if (prepare_batch() != FLS_OK) return 0;
        size_t nb = 0;
        const size_t used_now = live_counts(nb);
        return size_t(image.flat_header(used_now, nb).total_bytes);
    }
    fls_status map_image_export(void* dst, size_t cap, int on_device) override {
        if (borrowed || replica_only || !dst) return FLS_ERR_STATE;
        const fls_status prc = prepare_batch();  // image current, the stream idle
        if (prc != FLS_OK) return prc;
        size_t nb = 0;
        const size_t used_now = live_counts(nb);
        IvoxImage::FlatHeader h = image.flat_header(used_now, nb);
        h.kind = unsigned(kind); h.is_first = is_first ? 1u : 0u; h.use_dense = h.have_bricks; h.resolution = ivox.resolution;
        if (cap < size_t(h.total_bytes)) return FLS_ERR_RANGE;
        image.export_flat(h, dst, on_device != 0, stream);
        return FLS_OK;
    }
    fls_status map_image_import(const void* src, size_t n, int on_device) override {
        if (borrowed || !src || n < sizeof(IvoxImage::FlatHeader)) return FLS_ERR_INVALID;
        IvoxImage::FlatHeader h;
        FLS_HIP(hipMemcpy(&h, src, sizeof(h), on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        if (!IvoxImage::flat_header_ok(h, n) || h.kind != unsigned(kind)) return FLS_ERR_INVALID;
        if (!(h.resolution > 0.f) || h.resolution != ivox.resolution) return FLS_ERR_INVALID;
        FLS_HIP(hipStreamSynchronize(stream));
        device_map = false;
        replica_only = false;
        ivox.clear();
        image_built = false; image_dirty = true;  // (an import that fails half-way leaves an empty map that rebuilds its image)
        use_dense = h.have_bricks != 0;  // (derived, not trusted: the query takes the brick path exactly when the image carries bricks)
        image.import_flat(h, src, on_device != 0, stream);
        // the contents are as untrusted as the header: no {begin, count} may leave the point array, no directory entry may name a missing brick
        d_counter_img.reserve(1);
        FLS_HIP(hipMemsetAsync(d_counter_img.p, 0, sizeof(unsigned), stream));
        const unsigned long long n_cells = image.have_bricks ? (unsigned long long)h.n_bricks_live * kBrickStride : 0ull;
        hipLaunchKernelGGL(ivox_image_validate_kernel, dim3(512), dim3(256), 0, stream, (const uint2*)image.d_cells.p, n_cells, (const HashEntry*)image.d_dir.p,
                           image.have_bricks ? (unsigned long long)h.dir_mask + 1ull : 0ull, unsigned(h.n_bricks_live), (const HashEntry*)image.d_table.p,
                           image.want_hash ? (unsigned long long)h.mask + 1ull : 0ull, (unsigned long long)h.used, d_counter_img.p);
        unsigned bad = 0;
        FLS_HIP(hipMemcpyAsync(&bad, d_counter_img.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        if (bad != 0u) return FLS_ERR_INVALID;  // (the handle is an empty map whose image is rebuilt from the empty mirror on its next use)
        is_first = h.is_first != 0;
        replica_only = true;
        image_built = true; image_dirty = false; rebuild_after_replay = false;
        nn_n = 0; have_final = false; nn_rows_current = true; spec_pending = false;
        return FLS_OK;
    }
    DevBuf<unsigned> d_counter_img;

    std::unique_ptr<fls_matcher> clone_for_lane() override {
        auto q = std::make_unique<P2PlaneIvoxMatcher>();
        q->kind = kind; q->p = p; q->device = device;
        if (q->init() != FLS_OK) return nullptr;
        q->borrowed = &image;
        return q;
    }
    fls_status prepare_batch() override {
        refresh_image();
        FLS_HIP(hipStreamSynchronize(stream));  // the image is complete before other streams read it
        return FLS_OK;
    }
    void tune_lane(fls_matcher& l) override {
        auto& q = static_cast<P2PlaneIvoxMatcher&>(l);
        q.use_dense = use_dense; q.variant = variant; q.balanced = balanced; q.xcd_chunk = xcd_chunk; q.prof_fit = prof_fit; q.ticket_shards = ticket_shards; q.plain_launch = plain_launch; q.nn_ids_mode = nn_ids_mode;
    }

    fls_status fitness(float max_range, float* score) override {
        if (!p.is_localization_mode) { *score = std::numeric_limits<float>::max(); return FLS_OK; }  // FloatNaN :226-228
        if (!have_fitness_grid || !have_final) return FLS_ERR_STATE;
        return fitness_score_device(*this, fitness_grid, scan, final_T, max_range, score);
    }

    int correspondences(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) override {
        download_nn();
        const size_t n = std::min(cap, nn_n);
        h_flag.resize(nn_n);
        if (nn_n) {
            FLS_HIP(hipMemcpyAsync(h_flag.data(), d_flag.p, nn_n, hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
        }
        for (size_t i = 0; i < n; ++i) {
            cnt[i] = h_cnt[i];
            for (int j = 0; j < 5; ++j) ids[i * 5 + j] = j < h_cnt[i] ? h_nn[i * 5 + j].id : -1;
            valid[i] = h_flag[i];
        }
        return int(n);
    }
    size_t map_size(int slot) const override {
        if (slot == 100) return n_incremental;    // introspection: image updates applied as scatter lists
        if (slot == 101) return n_full_rebuilds;  //                ... as full re-flatten + upload
        if (slot == 103) return n_device_updates;  //                ... by the device-side AddPoints
        if (slot == 104) return n_host_fallbacks;  //                batches the device refused (replayed on the host)
        if (slot == 122) return n_fused_updates;    //                ... of which in the one-launch form
        if (slot == 124) return n_speculative;      //                chains queued speculatively behind the iterations
        if (slot == 125) return n_speculative_skipped;  //            ... that the device skipped (the Match needed more iterations / did not converge)
        if (slot == 123) return n_short_updates;    //                ... of which in the short chain (five launches behind the decision)
        if (slot == 117) return n_device_evictions;  //              voxels evicted inside device batches
        if (slot == 126) return n_device_recreated;  //              ... of which re-created by a later point of the same batch (eviction-order conflicts resolved on the device)
        if (slot == 119) return n_refused_conflict;  //              refusals by reason: eviction order conflict / point array full / point outside the window
        if (slot == 132) return image.budget_exceeded ? 1u : 0u;  // the brick pool went over FLS_IVOX_BRICK_BUDGET_MB: hash-table image, host AddPoints
        if (slot == 120) return n_refused_full;
        if (slot == 121) return n_refused_outside;
        if (slot == 102) return device_map ? dev_n_alive : ivox.n_alive;     // occupied voxels
        return device_map ? dev_n_points : ivox.n_points;
    }
};

}  // namespace fls
