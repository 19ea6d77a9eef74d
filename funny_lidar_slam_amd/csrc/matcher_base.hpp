// matcher_base.hpp -- common host state of one registration handle (stream, device
// Gauss-Newton state, wave partials, profiling events) and the scan upload helper.
#pragma once
#include "host_maps.hpp"
#include <memory>

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
    std::vector<hipEvent_t> ev;  // 2 per possible iteration
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

    virtual ~fls_matcher() {
        for (auto e : ev) (void)hipEventDestroy(e);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (mb_host) (void)hipHostFree(mb_host);
    }
    virtual fls_status add_cloud(const float* c0, size_t n0, const float* c1, size_t n1, int stride) = 0;
    virtual fls_status scan_upload(const float* s0, size_t n0, const float* s1, size_t n1, int stride) = 0;
    virtual fls_status match_resident(double* T, int update_map, fls_stats* out) = 0;
    virtual fls_status fitness(float max_range, float* score) = 0;
    virtual int correspondences(int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) = 0;
    virtual size_t map_size(int slot) const = 0;

    void init_common() {
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
        if (const char* e = std::getenv("FLS_GN_CHUNK")) {  // experiments: "first,next"
            int a = 0, b = 0;
            if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { chunk_first = a; chunk_next = b; }
        }
    }
    void ensure_events(int iters) {
        while ((int)ev.size() < 2 * iters) {
            hipEvent_t e;
            FLS_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
    }
    // reset the device Gauss-Newton state with the initial guess
    void push_state(const double* T) {
        fls::GnState* hs = h_state.p;
        std::memset(hs, 0, offsetof(fls::GnState, log_T));
        std::memcpy(hs->T, T, sizeof(double) * 16);
        FLS_HIP(hipMemcpyAsync(d_state.p, hs, offsetof(fls::GnState, log_T), hipMemcpyHostToDevice, stream));
        if (count_traffic) FLS_HIP(hipMemsetAsync(d_tc.p, 0, sizeof(fls::TrafficCounters), stream));
    }
    // Gauss-Newton launch loop.  Iterations are enqueued in chunks (first 4, then 3 at a time): the device
    // decides convergence, kernels of an already converged Match exit at once, and the host only peeks at the
    // `done` flag between chunks -- so a typical 3-5 iteration Match costs one synchronisation and at most a
    // few microseconds of dead launches instead of (max_iterations - used) x 2 of them.
    int chunk_first = 4, chunk_next = 3;
    template <class F>
    void run_gn_loop(int iters, F&& launch_iteration) {
        int it = 0, chunk = std::min(iters, chunk_first);
        while (it < iters) {
            const int end = std::min(iters, it + chunk);
            for (; it < end; ++it) launch_iteration(it);
            if (it >= iters) break;
            FLS_HIP(hipMemcpyAsync(h_state.p, d_state.p, offsetof(fls::GnState, log_T), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
            if (h_state.p->done) break;
            chunk = chunk_next;
        }
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
    // account the hipEvent-bracketed launches of the executed iterations
    void account_profile(int iters, size_t points_per_iter) {
        if (!profiling) return;
        for (int i = 0; i < iters && 2 * i + 1 < (int)ev.size(); ++i) {
            float ms = 0.f;
            FLS_HIP(hipEventSynchronize(ev[2 * i + 1]));
            FLS_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            prof_ms += ms;
            prof_launches += 1;
        }
        prof_point_iters += uint64_t(iters) * points_per_iter;
    }
    // read the state back (one synchronisation per Match) and account the profiled launches
    void pull_state(size_t points_per_iter) {
        FLS_HIP(hipMemcpyAsync(h_state.p, d_state.p, sizeof(fls::GnState), hipMemcpyDeviceToHost, stream));
        if (count_traffic) FLS_HIP(hipMemcpyAsync(h_tc.p, d_tc.p, sizeof(fls::TrafficCounters), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        log_n = std::min(h_state.p->iter, fls::kMaxIter);
        if (profiling) {
            const int iters = h_state.p->iter;
            for (int i = 0; i < iters && 2 * i + 1 < (int)ev.size(); ++i) {
                float ms = 0.f;
                FLS_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
                prof_ms += ms;
                prof_launches += 1;
            }
            prof_point_iters += uint64_t(iters) * points_per_iter;
        }
        if (count_traffic) last_tc = *h_tc.p;
    }
};

namespace fls {

// SoA device copy of a source cloud
struct DevScan {
    DevBuf<float> x, y, z;
    PinnedBuf<float> stage;
    size_t n = 0;
    std::vector<PtI> host;  // kept for map updates / fitness
    void upload(const std::vector<PtI>& c, hipStream_t s) {
        host = c;
        n = c.size();
        if (n == 0) return;
        x.reserve(n); y.reserve(n); z.reserve(n);
        stage.reserve(3 * n);
        for (size_t i = 0; i < n; ++i) { stage.p[i] = c[i].x; stage.p[n + i] = c[i].y; stage.p[2 * n + i] = c[i].z; }
        FLS_HIP(hipMemcpyAsync(x.p, stage.p, n * sizeof(float), hipMemcpyHostToDevice, s));
        FLS_HIP(hipMemcpyAsync(y.p, stage.p + n, n * sizeof(float), hipMemcpyHostToDevice, s));
        FLS_HIP(hipMemcpyAsync(z.p, stage.p + 2 * n, n * sizeof(float), hipMemcpyHostToDevice, s));
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
