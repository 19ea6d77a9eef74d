// matcher_base.hpp -- common host state of one registration handle (stream, device
// Gauss-Newton state, wave partials, profiling events) and the scan upload helper.
#pragma once
#include "host_maps.hpp"
#include <memory>
#include <thread>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

struct fls_matcher {
    fls_kind kind;
    fls_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    fls_stats stats{};

    fls::DevBuf<fls::GnState> d_state;
    fls::PinnedBuf<fls::GnState> h_state;
    fls::DevBuf<double> d_partials_a, d_partials_b;
    fls::DevBuf<fls::TrafficCounters> d_tc;
    fls::PinnedBuf<fls::TrafficCounters> h_tc;

    // profiling (fls_set_profiling)
    bool profiling = false;      // hipEvents around every correspondence launch
    bool count_traffic = false;  // run the <COUNT=true> kernel variants (device traffic counters)
    // hipEvent pairs around the correspondence launches: a ring of kEvRing Matches so that nothing is resolved
    // (hipEventSynchronize / ElapsedTime) on the timed path; fls_get_kernel_time settles the pending pairs
    static constexpr int kEvRing = 64;
    std::vector<hipEvent_t> ev_pool;   // kEvRing x 2 x kMaxIter, created lazily
    hipEvent_t* ev = nullptr;          // the current Match's slice: ev[2 * it], ev[2 * it + 1]
    int ev_slot = -1;
    struct PendingEv { int slot, iters; };
    std::vector<PendingEv> ev_pending;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    uint64_t prof_point_iters = 0;
    fls::TrafficCounters last_tc{0, 0, 0};

    // iteration log of the last Match
    int log_n = 0;
    bool log_stale = false;  // mailbox path: the full GnState (with the per-iteration log) is fetched on demand

    // result mailbox (host-mapped pinned memory, see device_common.hpp)
    fls::Mailbox* mb_host = nullptr;
    fls::Mailbox* mb_dev = nullptr;
    unsigned match_id = 0;
    // what the tail kernels get: max_iterations << 24 | exact-solver flag << 23 | match id (mailbox_publish, device_common.hpp)
    bool tail_exact = false;  // FLS_TAIL_EXACT=1: every 6x6 system through the Eigen-arithmetic solver (no LDL^T fast path)
    unsigned launch_word() const {
        return (match_id & 0x7fffffu) | (tail_exact ? (1u << 23) : 0u) | (std::min<unsigned>(p.max_iterations, 255u) << 24);
    }

    virtual ~fls_matcher() {
        for (auto e : ev_pool) if (e) (void)hipEventDestroy(e);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (mb_host) (void)hipHostFree(mb_host);
    }
    virtual fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) = 0;
    virtual fls_status scan_upload(const float* s0, size_t n0, const float* s1, size_t n1, int stride) = 0;
    // fls_match's upload: the Match follows at once, so a kind may leave the scan in its pinned staging buffer and let the first
    // iteration's kernels read it from there (default: the ordinary upload)
    virtual fls_status scan_upload_for_match(const float* s0, size_t n0, const float* s1, size_t n1, int stride) { return scan_upload(s0, n0, s1, n1, stride); }
    // fls_scan_upload_raw: kinds whose Match filters its source (ICP, NDT) keep the RAW scan resident and filter inside match_resident
    virtual fls_status scan_upload_raw(const float* s0, size_t n0, const float* s1, size_t n1, int stride) { return scan_upload(s0, n0, s1, n1, stride); }
    virtual fls_status match_resident(double* T, int update_map, fls_stats* out) = 0;
    virtual fls_status fitness(float max_range, float* score) = 0;
    virtual int correspondences(int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) = 0;
    virtual size_t map_size(int slot) const = 0;
    virtual size_t map_export(void*, size_t) { return 0; }                       // 0: this kind has no exportable image
    virtual fls_status map_import(const void*, size_t) { return FLS_ERR_STATE; }
    // the DEVICE image as one flat buffer (fls_map_image_*): 0 / FLS_ERR_STATE: this kind has none
    virtual size_t map_image_bytes() { return 0; }
    virtual fls_status map_image_export(void*, size_t, int /*buffer on this handle's device*/) { return FLS_ERR_STATE; }
    virtual fls_status map_image_import(const void*, size_t, int) { return FLS_ERR_STATE; }
    // fls_replicas_*: this handle becomes a READ-ONLY copy of `owner`'s device map image (device to device, no host mirror): it serves
    // fls_match_batch (and fls_match with update_map == 0) and nothing else until fls_map_import makes it an ordinary handle again (fls_add_cloud_to_local_map is refused: FLS_ERR_STATE)
    virtual bool can_replicate() const { return false; }
    virtual fls_status replicate_from(fls_matcher&) { return FLS_ERR_STATE; }
    // state a FRESH reference matcher would not have (e.g. nearest_points_ of an earlier Match): cleared per batch job
    virtual void reset_job_state() {}
    // Batch of independent registrations against the CURRENT map (BASELINE configs[4]): every job is what a
    // fresh reference process holding this map would compute for Match(scan_j, T_j) -- no map update, no state carried
    // from job to job.  With lanes > 1 the jobs run on `lanes` clones of this handle (own stream, own
    // Gauss-Newton state, mailbox and per-point buffers; one host thread each, spinning on its own mailbox) that READ
    // this handle's resident map: while one job runs its single-workgroup tail or an under-occupied fit kernel, the
    // correspondence kernels of the other jobs fill the machine.  lanes <= 1: one clone, jobs one after the other.
    std::vector<std::unique_ptr<fls_matcher>> lanes;
    bool is_lane = false;
    virtual std::unique_ptr<fls_matcher> clone_for_lane() { return nullptr; }  // same kind, borrowing this handle's map
    virtual fls_status prepare_batch() { return FLS_OK; }                     // map image current and complete on the device
    virtual void tune_lane(fls_matcher&) {}                                    // copy run-time switches to a lane
    fls_status match_batch(size_t n_jobs, const float* const* s0, const size_t* n0, const float* const* s1, const size_t* n1, int stride,
                           double* T, fls_stats* st, int32_t* status, int n_lanes) {
        if (is_lane) return FLS_ERR_STATE;
        if (status) for (size_t j = 0; j < n_jobs; ++j) status[j] = FLS_SKIPPED;  // overwritten by every job that runs
        if (n_jobs == 0) return FLS_OK;
        size_t L = std::min(size_t(std::max(1, std::min(n_lanes, 16))), n_jobs);
        // Jobs never run on the owner itself: its Match state (nearest_points_, keyframe gate, resident scan, final pose)
        // belongs to the SLAM thread's next fls_match.  lanes <= 1 (or a single job) = one lane clone, back to back.
        const fls_status prc = prepare_batch();
        if (prc != FLS_OK) return prc;
        while (lanes.size() < L) {
            std::unique_ptr<fls_matcher> q = clone_for_lane();
            if (!q) break;
            q->is_lane = true;
            lanes.push_back(std::move(q));
        }
        if (lanes.empty()) return FLS_ERR_NOMEM;  // the clone could not be set up
        L = std::min(L, lanes.size());
        std::vector<fls_status> lane_rc(L, FLS_OK);
        std::vector<std::thread> th;
        struct JoinAll {  // a joinable std::thread must never be destroyed: also when starting a later lane throws
            std::vector<std::thread>& t;
            ~JoinAll() { for (auto& x : t) if (x.joinable()) x.join(); }
        } join_all{th};
        for (size_t l = 0; l < L; ++l) {
            fls_matcher* q = lanes[l].get();
            tune_lane(*q);
            q->expect_iters = expect_iters;
            th.emplace_back([=, &lane_rc]() {
                try {
                    FLS_HIP(hipSetDevice(q->device));
                    for (size_t j = l; j < n_jobs; j += L) {
                        q->reset_job_state();
                        fls_status rc = q->scan_upload(s0[j], n0[j], s1 ? s1[j] : nullptr, n1 ? n1[j] : 0, stride);
                        if (rc == FLS_OK) rc = q->match_resident(T + 16 * j, 0, st ? &st[j] : nullptr);
                        if (status) status[j] = int32_t(rc);
                        if (rc < 0) { lane_rc[l] = rc; return; }
                    }
                } catch (const fls::HipError& e) {
                    std::fprintf(stderr, "[fls_reg] batch lane %zu: %s\n", l, e.what());
                    lane_rc[l] = FLS_ERR_DEVICE;
                } catch (const std::bad_alloc&) {
                    lane_rc[l] = FLS_ERR_NOMEM;
                } catch (...) {
                    lane_rc[l] = FLS_ERR_INVALID;
                }
            });
        }
        for (auto& t : th) t.join();  // (join_all is the exception path)
        for (const fls_status rc : lane_rc) if (rc < 0) return rc;
        return FLS_OK;
    }

    void init_common() {
        if (const char* e = std::getenv("FLS_TAIL_EXACT")) tail_exact = std::atoi(e) != 0;
        FLS_HIP(hipSetDevice(device));
        FLS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        d_state.reserve(1);
        h_state.reserve(1);
        d_tc.reserve(1);
        h_tc.reserve(1);
        FLS_HIP(hipMemsetAsync(d_tc.p, 0, sizeof(fls::TrafficCounters), stream));
        FLS_HIP(hipHostMalloc((void**)&mb_host, sizeof(fls::Mailbox), hipHostMallocMapped));
        std::memset(mb_host, 0, sizeof(fls::Mailbox));
        mb_host->seq = 0u;
        FLS_HIP(hipHostGetDevicePointer((void**)&mb_dev, mb_host, 0));
    }
    void ensure_events(int iters) {
        if (ev_pool.empty()) {
            // the whole ring at once, on the first profiled Match (bench.py makes that an untimed one): creating a slot's events lazily
            // put ~20 hipEventCreate calls (30+ us) inside every bracketed step that met a fresh slot
            ev_pool.assign(size_t(kEvRing) * 2 * fls::kMaxIter, nullptr);
            for (hipEvent_t& e : ev_pool) FLS_HIP(hipEventCreate(&e));
        }
        ev_slot = (ev_slot + 1) % kEvRing;
        for (const PendingEv& pe : ev_pending)
            if (pe.slot == ev_slot) { settle_events(); break; }  // the ring wrapped around
        ev = ev_pool.data() + size_t(ev_slot) * 2 * fls::kMaxIter;
        for (int i = 0; i < 2 * iters; ++i)
            if (!ev[i]) FLS_HIP(hipEventCreate(&ev[i]));
    }
    void settle_events() {
        for (const PendingEv& pe : ev_pending) {
            hipEvent_t* e = ev_pool.data() + size_t(pe.slot) * 2 * fls::kMaxIter;
            for (int i = 0; i < pe.iters; ++i) {
                float ms = 0.f;
                FLS_HIP(hipEventSynchronize(e[2 * i + 1]));
                FLS_HIP(hipEventElapsedTime(&ms, e[2 * i], e[2 * i + 1]));
                prof_ms += ms;
                prof_launches += 1;
            }
        }
        ev_pending.clear();
    }
    // Spin on the mailbox until the Gauss-Newton tail of iteration `target` (or an earlier one that hit the stop
    // rule) has published its result.  Returns the published word.  Falls back to the stream state every
    // few thousand polls so that a faulted kernel cannot hang the host.
    unsigned wait_mailbox(int target_iter) {
        const unsigned want = match_id & 0x7fffffu;
        for (unsigned long long spin = 1;; ++spin) {
            const unsigned s = __atomic_load_n(&mb_host->seq, __ATOMIC_ACQUIRE);
            if ((s >> 9) == want && (((s >> 8) & 1u) || int(s & 0xffu) >= target_iter)) return s;
            if ((spin & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) {  // everything drained: the word is final
                    return __atomic_load_n(&mb_host->seq, __ATOMIC_ACQUIRE);
                }
                if (q != hipErrorNotReady) FLS_HIP(q);
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    // fetch the full device state (iteration log) after a mailbox-path Match
    void refresh_log() {
        if (!log_stale) return;
        FLS_HIP(hipMemcpyAsync(h_state.p, d_state.p, sizeof(fls::GnState), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        log_n = std::min(h_state.p->iter, fls::kMaxIter);
        log_stale = false;
    }
    // remember the hipEvent-bracketed launches of the executed iterations (settled lazily)
    void account_profile(int iters, size_t points_per_iter) {
        if (!profiling) return;
        ev_pending.push_back(PendingEv{ev_slot, std::min(iters, fls::kMaxIter)});
        prof_point_iters += uint64_t(iters) * points_per_iter;
    }
    // Gauss-Newton launch loop shared by every kind.  Iterations are enqueued in chunks sized by the previous
    // Match (steady-state SLAM needs about the same number every scan); the device decides convergence, kernels
    // of a finished Match exit at once, and the host learns the outcome from the mailbox without a blocking
    // synchronisation.  launch(it, first) enqueues one iteration; returns the published mailbox word.
    int expect_iters = 4;
    template <class F>
    unsigned run_mailbox_loop(int iters, size_t points_per_iter, F&& launch) {
        return run_mailbox_loop(iters, points_per_iter, launch, [](int) {});
    }
    // after_chunk(launched): called after every chunk of iterations has been queued, before the host waits (speculative work behind the chunk)
    template <class F, class G>
    unsigned run_mailbox_loop(int iters, size_t points_per_iter, F&& launch, G&& after_chunk) {
        match_id = (match_id + 1) & 0x7fffffu;
        if (profiling) ensure_events(iters);
        if (count_traffic) FLS_HIP(hipMemsetAsync(d_tc.p, 0, sizeof(fls::TrafficCounters), stream));
        int launched = 0;
        unsigned word = 0;
        int chunk = std::max(1, std::min(iters, expect_iters));
        for (;;) {
            const int end = std::min(iters, launched + chunk);
            for (int it = launched; it < end; ++it) launch(it, it == 0 ? 1 : 0);
            launched = end;
            after_chunk(launched);
            FLS_HIP(hipGetLastError());
            word = wait_mailbox(launched);
            if (((word >> 8) & 1u) || launched >= iters) break;
            chunk = 2;
        }
        const int used = int(word & 0xffu);
        expect_iters = std::max(2, used);
        log_stale = true;
        log_n = std::min(used, fls::kMaxIter);
        account_profile(used, points_per_iter);
        if (count_traffic) {
            FLS_HIP(hipMemcpyAsync(h_tc.p, d_tc.p, sizeof(fls::TrafficCounters), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
            last_tc = *h_tc.p;
        }
        return word;
    }
};

namespace fls {

// SoA device copy of a source cloud
struct DevScan {
    struct View { float* p = nullptr; };
    DevBuf<float> xyz;  // x[n] | y[n] | z[n] in one allocation: one host-to-device copy per upload
    View x, y, z;
    PinnedBuf<float> stage;
    size_t n = 0;
    std::vector<PtI> host;  // kept for map updates / fitness
    void push(hipStream_t s, int fields = 3) {
        xyz.reserve(size_t(fields) * n);
        x.p = xyz.p; y.p = xyz.p + n; z.p = xyz.p + 2 * n;
        FLS_HIP(hipMemcpyAsync(xyz.p, stage.p, size_t(fields) * n * sizeof(float), hipMemcpyHostToDevice, s));
    }
    // straight from the caller's strided AoS into the pinned SoA staging buffer x[n] | y[n] | z[n] | intensity[n] (no
    // temporary cloud, no per-point host copy).  The first 3 n floats go to the device; the staged intensities stay valid
    // until the next upload, which is all a map update of THIS resident scan needs (fls_match == fls_scan_upload +
    // fls_match_resident for either value of update_map).
    void upload_raw(const float* p, size_t count, int stride, hipStream_t s, bool intensity_too = false) {
        stage_raw(p, count, stride);
        if (n) push(s, intensity_too ? 4 : 3);  // the device VoxelGrid averages the intensity as well
    }
    // device-side address of the staging buffer (pinned host memory is mapped into the device's address space)
    const float* stage_dev() const {
        void* d = nullptr;
        FLS_HIP(hipHostGetDevicePointer(&d, stage.p, 0));
        return static_cast<const float*>(d);
    }
    // the device allocation x | y | z of the staged scan, without the copy (a kernel fills it: ivox_knn_kernel's first launch)
    void reserve_device() {
        xyz.reserve(size_t(3) * n);
        x.p = xyz.p; y.p = xyz.p + n; z.p = xyz.p + 2 * n;
    }
    void stage_raw(const float* p, size_t count, int stride) {
        n = count;
        host.clear();
        if (n == 0) return;
        stage.reserve(4 * n);
        float* sx = stage.p; float* sy = stage.p + n; float* sz = stage.p + 2 * n; float* si = stage.p + 3 * n;
        size_t i = 0;
#if defined(__SSE__)
        if (stride == 4) {  // packed xyzi rows: 4 x 4 transposes (45 -> ~12 us for 115,200 points)
            for (; i + 4 <= n; i += 4) {
                __m128 r0 = _mm_loadu_ps(p + 4 * i), r1 = _mm_loadu_ps(p + 4 * i + 4), r2 = _mm_loadu_ps(p + 4 * i + 8), r3 = _mm_loadu_ps(p + 4 * i + 12);
                _MM_TRANSPOSE4_PS(r0, r1, r2, r3);
                _mm_storeu_ps(sx + i, r0); _mm_storeu_ps(sy + i, r1); _mm_storeu_ps(sz + i, r2); _mm_storeu_ps(si + i, r3);
            }
        } else if (stride == 3) {  // packed xyz: three vectors hold four points
            const __m128 zero = _mm_setzero_ps();
            for (; i + 4 <= n; i += 4) {
                const __m128 a = _mm_loadu_ps(p + 3 * i), b = _mm_loadu_ps(p + 3 * i + 4), c = _mm_loadu_ps(p + 3 * i + 8);
                const __m128 m2 = _mm_shuffle_ps(b, c, _MM_SHUFFLE(2, 1, 3, 2));  // b2 b3 c1 c2
                const __m128 m1 = _mm_shuffle_ps(a, b, _MM_SHUFFLE(1, 0, 2, 1));  // a1 a2 b0 b1
                _mm_storeu_ps(sx + i, _mm_shuffle_ps(a, m2, _MM_SHUFFLE(2, 0, 3, 0)));   // a0 a3 b2 c1
                _mm_storeu_ps(sy + i, _mm_shuffle_ps(m1, m2, _MM_SHUFFLE(3, 1, 2, 0)));  // a1 b0 b3 c2
                _mm_storeu_ps(sz + i, _mm_shuffle_ps(m1, c, _MM_SHUFFLE(3, 0, 3, 1)));   // a2 b1 c0 c3
                _mm_storeu_ps(si + i, zero);
            }
        } else if (stride == 8) {  // pcl::PointXYZI: {x, y, z, pad | intensity, pad, pad, pad}
            for (; i + 4 <= n; i += 4) {
                __m128 r0 = _mm_loadu_ps(p + 8 * i), r1 = _mm_loadu_ps(p + 8 * i + 8), r2 = _mm_loadu_ps(p + 8 * i + 16), r3 = _mm_loadu_ps(p + 8 * i + 24);
                _MM_TRANSPOSE4_PS(r0, r1, r2, r3);
                _mm_storeu_ps(sx + i, r0); _mm_storeu_ps(sy + i, r1); _mm_storeu_ps(sz + i, r2);
                si[i] = p[8 * i + 4]; si[i + 1] = p[8 * i + 12]; si[i + 2] = p[8 * i + 20]; si[i + 3] = p[8 * i + 28];
            }
        }
#endif
        for (; i < n; ++i) {
            const float* q = p + i * stride;
            sx[i] = q[0]; sy[i] = q[1]; sz[i] = q[2]; si[i] = intensity_of(q, stride);
        }
    }
    float staged_intensity(size_t i) const { return stage.p[3 * n + i]; }
    void upload(const std::vector<PtI>& c, hipStream_t s) {
        host = c;
        n = c.size();
        if (n == 0) return;
        stage.reserve(3 * n);
        for (size_t i = 0; i < n; ++i) { stage.p[i] = c[i].x; stage.p[n + i] = c[i].y; stage.p[2 * n + i] = c[i].z; }
        push(s);
    }
};

inline fls_status check_common(const fls_params& p) {
    if (p.struct_size != sizeof(fls_params)) return FLS_ERR_INVALID;
    if (p.max_iterations == 0 || p.max_iterations > (uint32_t)kMaxIter) return FLS_ERR_INVALID;
    return FLS_OK;
}
inline bool unset_d(double v) { return v == std::numeric_limits<double>::max() || !(v == v); }
inline bool unset_f(float v) { return v == std::numeric_limits<float>::max() || !(v == v); }

}  // namespace fls
