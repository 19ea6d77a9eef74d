// fls_reg.hip -- libfls_reg.so: C ABI (include/fls_reg.h) over the gfx950 registration back-end.
// Single translation unit: device kernels (kernels_*.hpp) + host matchers (matcher*.hpp).
// Built by csrc/Makefile:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// There is no CPU fallback: without a gfx950 device fls_create fails with FLS_ERR_DEVICE.
#include <chrono>
#include "matcher_p2plane_ivox.hpp"
#include "matchers_kd.hpp"
#include "matcher_ndt.hpp"
#include "features_host.hpp"
#include "loop_closure.hpp"
#include "replicas.hpp"
#include <new>

using namespace fls;

namespace {

int gfx950_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

template <typename F>
fls_status guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        std::fprintf(stderr, "[fls_reg] %s\n", e.what());
        return FLS_ERR_DEVICE;
    } catch (const std::bad_alloc&) {
        return FLS_ERR_NOMEM;
    } catch (...) {
        return FLS_ERR_INVALID;
    }
}

}  // namespace

extern "C" {

int fls_abi_version(void) { return FLS_ABI_VERSION; }
int fls_abi_revision(void) { return FLS_ABI_REVISION; }
int fls_device_count(void) { return gfx950_device_count(); }

const char* fls_status_string(int s) {
    switch (s) {
        case FLS_OK: return "ok (Match returned true)";
        case FLS_NOT_CONVERGED: return "not converged (Match returned false)";
        case FLS_SKIPPED: return "batch job not run";
        case FLS_ERR_INVALID: return "invalid argument or unset parameter";
        case FLS_ERR_DEVICE: return "HIP error or no gfx950 device";
        case FLS_ERR_RANGE: return "coordinate outside the voxel key range";
        case FLS_ERR_NOMEM: return "out of memory";
        case FLS_ERR_STATE: return "call order violated";
        default: return "unknown status";
    }
}

fls_status fls_create(fls_kind kind, const fls_params* params, int device_id, fls_handle* out) {
    if (!out) return FLS_ERR_INVALID;
    *out = nullptr;
    if (!params) return FLS_ERR_INVALID;
    const fls_status pc = check_common(*params);
    if (pc != FLS_OK) return pc;
    return guarded([&]() -> fls_status {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return FLS_ERR_DEVICE;
        hipDeviceProp_t prop;
        FLS_HIP(hipGetDeviceProperties(&prop, device_id));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            std::fprintf(stderr, "[fls_reg] device %d is %s; this library is built for gfx950 only\n", device_id, prop.gcnArchName);
            return FLS_ERR_DEVICE;
        }
        std::unique_ptr<fls_matcher> m;
        fls_status rc = FLS_ERR_INVALID;
        switch (kind) {
            case FLS_P2PLANE_IVOX: { auto* q = new P2PlaneIvoxMatcher(); m.reset(q); q->kind = kind; q->p = *params; q->device = device_id; rc = q->init(); break; }
            case FLS_ICP_OPTIMIZED: { auto* q = new IcpMatcher(); m.reset(q); q->kind = kind; q->p = *params; q->device = device_id; rc = q->init(); break; }
            case FLS_INCREMENTAL_NDT: { auto* q = new NdtMatcher(); m.reset(q); q->kind = kind; q->p = *params; q->device = device_id; rc = q->init(); break; }
            case FLS_LOAM_FULL: { auto* q = new LoamFullMatcher(); m.reset(q); q->kind = kind; q->p = *params; q->device = device_id; rc = q->init(); break; }
            case FLS_P2PLANE_KDTREE: { auto* q = new P2PlaneKdMatcher(); m.reset(q); q->kind = kind; q->p = *params; q->device = device_id; rc = q->init(); break; }
            default: return FLS_ERR_INVALID;
        }
        if (rc != FLS_OK) return rc;
        *out = m.release();
        return FLS_OK;
    });
}

void fls_destroy(fls_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->lanes.clear();  // the batch lanes read this handle's map: they go first
    delete h;
}

fls_status fls_add_cloud_to_local_map(fls_handle h, const float* c0, size_t n0, const float* c1, size_t n1, int stride) {
    if (!h || (!c0 && n0) || stride < 3) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->add_cloud(c0, n0, c1, n1, stride);
    });
}

fls_status fls_scan_upload(fls_handle h, const float* s0, size_t n0, const float* s1, size_t n1, int stride) {
    if (!h || (!s0 && n0) || stride < 3) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        const fls_status rc = h->scan_upload(s0, n0, s1, n1, stride);
        FLS_HIP(hipStreamSynchronize(h->stream));
        return rc;
    });
}

fls_status fls_scan_upload_raw(fls_handle h, const float* s0, size_t n0, const float* s1, size_t n1, int stride) {
    if (!h || (!s0 && n0) || stride < 3) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        const fls_status rc = h->scan_upload_raw(s0, n0, s1, n1, stride);
        FLS_HIP(hipStreamSynchronize(h->stream));
        return rc;
    });
}

fls_status fls_match_resident(fls_handle h, double T[16], int update_map, fls_stats* stats) {
    if (!h || !T) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->match_resident(T, update_map, stats);
    });
}

fls_status fls_match(fls_handle h, const float* s0, size_t n0, const float* s1, size_t n1, int stride, double T[16], int update_map,
                     fls_stats* stats) {
    if (!h || !T || (!s0 && n0) || stride < 3) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        const fls_status rc = h->scan_upload_for_match(s0, n0, s1, n1, stride);
        if (rc != FLS_OK) return rc;
        return h->match_resident(T, update_map, stats);
    });
}

size_t fls_map_image_bytes(fls_handle h) {
    if (!h) return 0;
    size_t n = 0;
    guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); n = h->map_image_bytes(); return FLS_OK; });
    return n;
}
fls_status fls_map_image_export(fls_handle h, void* dst, size_t cap_bytes, int dst_on_device) {
    if (!h || !dst) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); return h->map_image_export(dst, cap_bytes, dst_on_device); });
}
fls_status fls_map_image_import(fls_handle h, const void* src, size_t n_bytes, int src_on_device) {
    if (!h || !src) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); return h->map_image_import(src, n_bytes, src_on_device); });
}

fls_status fls_match_batch(fls_handle h, size_t n_jobs, const float* const* src0, const size_t* n0, const float* const* src1,
                           const size_t* n1, int stride, double* T, fls_stats* stats, int32_t* status, int lanes) {
    if (!h || stride < 3 || (n_jobs && (!src0 || !n0 || !T))) return FLS_ERR_INVALID;
    if ((src1 == nullptr) != (n1 == nullptr)) return FLS_ERR_INVALID;
    for (size_t j = 0; j < n_jobs; ++j)
        if (!src0[j] && n0[j]) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->match_batch(n_jobs, src0, n0, src1, n1, stride, T, stats, status, lanes);
    });
}

size_t fls_map_export(fls_handle h, void* blob, size_t cap) {
    if (!h) return 0;
    size_t n = 0;
    (void)guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); n = h->map_export(blob, cap); return FLS_OK; });
    return n;
}

fls_status fls_map_import(fls_handle h, const void* blob, size_t n) {
    if (!h || !blob) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); return h->map_import(blob, n); });
}

fls_status fls_replicas_create(fls_handle owner, const int* device_ids, int n_devices, fls_replicas_handle* out) {
    if (!out) return FLS_ERR_INVALID;
    *out = nullptr;
    if (!owner || !device_ids || n_devices <= 0 || n_devices > 64) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FLS_ERR_DEVICE;
        for (int i = 0; i < n_devices; ++i)
            if (device_ids[i] < 0 || device_ids[i] >= n) return FLS_ERR_DEVICE;
        std::unique_ptr<fls_replicas> r(new fls_replicas());
        r->owner = owner;
        bool owner_used = false;
        for (int i = 0; i < n_devices; ++i) {
            r->devices.push_back(device_ids[i]);
            r->import_ms.push_back(0.0);
            if (device_ids[i] == owner->device && !owner_used) {
                owner_used = true;
                r->handles.push_back(owner);
                r->owned.push_back(0);
                continue;
            }
            fls_handle h = nullptr;
            const fls_status rc = fls_create(owner->kind, &owner->p, device_ids[i], &h);
            if (rc < 0) return rc;  // (the set's destructor releases what was created so far)
            r->handles.push_back(h);
            r->owned.push_back(1);
        }
        const fls_status rc = r->refresh();
        if (rc < 0) return rc;
        *out = r.release();
        return FLS_OK;
    });
}

fls_status fls_replicas_refresh(fls_replicas_handle r) {
    if (!r) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status { return r->refresh(); });
}

fls_status fls_replicas_match_batch(fls_replicas_handle r, size_t n_jobs, const float* const* src0, const size_t* n0, const float* const* src1,
                                    const size_t* n1, int stride, double* T, fls_stats* stats, int32_t* status, int lanes) {
    if (!r || stride < 3 || (n_jobs && (!src0 || !n0 || !T))) return FLS_ERR_INVALID;
    if ((src1 == nullptr) != (n1 == nullptr)) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status { return r->match_batch(n_jobs, src0, n0, src1, n1, stride, T, stats, status, lanes); });
}

int fls_replicas_import_ms(fls_replicas_handle r, double* ms, int cap) {
    if (!r) return 0;
    const int n = int(r->import_ms.size());
    for (int i = 0; i < std::min(n, cap); ++i)
        if (ms) ms[i] = r->import_ms[size_t(i)];
    return n;
}

void fls_replicas_destroy(fls_replicas_handle r) { delete r; }

fls_status fls_get_fitness_score(fls_handle h, float max_range, float* score) {
    if (!h || !score) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->fitness(max_range, score);
    });
}

int fls_get_iteration_log(fls_handle h, double* T_iters, int32_t* n_valid, double* sum_res, int cap) {
    if (!h || !h->h_state.p) return 0;
    if (guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); h->refresh_log(); return FLS_OK; }) != FLS_OK) return 0;
    const GnState& s = *h->h_state.p;
    const int n = std::min(cap, h->log_n);
    for (int i = 0; i < n; ++i) {
        if (T_iters) std::memcpy(T_iters + 16 * i, s.log_T[i], sizeof(double) * 16);
        if (n_valid) n_valid[i] = s.log_nv[i];
        if (sum_res) sum_res[i] = s.log_res[i];
    }
    return h->log_n;
}

int fls_get_correspondences(fls_handle h, int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) {
    if (!h || !ids || !cnt || !valid) return -1;
    int n = -1;
    const fls_status rc = guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        n = h->correspondences(slot, ids, cnt, valid, cap);
        return FLS_OK;
    });
    return rc == FLS_OK ? n : -1;
}

size_t fls_map_size(fls_handle h, int slot) { return h ? h->map_size(slot) : 0; }

fls_status fls_set_profiling(fls_handle h, int enable) {
    if (!h) return FLS_ERR_INVALID;
    h->profiling = (enable & 1) != 0;       // accumulators are only reset by fls_get_kernel_time, so the flag can be
    h->count_traffic = (enable & 2) != 0;   // toggled per Match (e.g. to bracket every n-th step of a timed region)
    return FLS_OK;
}

fls_status fls_get_kernel_time(fls_handle h, double* ms_total, int64_t* launches, uint64_t* point_iters) {
    if (!h) return FLS_ERR_INVALID;
    const fls_status rc = guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); h->settle_events(); return FLS_OK; });
    if (rc != FLS_OK) return rc;
    if (ms_total) *ms_total = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (point_iters) *point_iters = h->prof_point_iters;
    h->prof_ms = 0.0;
    h->prof_launches = 0;
    h->prof_point_iters = 0;
    return FLS_OK;
}

fls_status fls_get_debug_stamps(fls_handle h, int64_t out[16]) {
    if (!h || !out || !h->h_state.p) return FLS_ERR_INVALID;
    (void)guarded([&]() -> fls_status { FLS_HIP(hipSetDevice(h->device)); h->refresh_log(); return FLS_OK; });
    for (int i = 0; i < 16; ++i) out[i] = h->h_state.p->dbg[i];
    return FLS_OK;
}

fls_status fls_debug_fullpiv_qr6(int device_id, const double* H, const double* g, int n, double* x) {
    if (!H || !g || !x || n < 0) return FLS_ERR_INVALID;
    if (n == 0) return FLS_OK;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(device_id));
        DevBuf<double> dH, dg, dx;
        dH.reserve(size_t(n) * 36); dg.reserve(size_t(n) * 6); dx.reserve(size_t(n) * 6);
        FLS_HIP(hipMemcpy(dH.p, H, size_t(n) * 36 * sizeof(double), hipMemcpyHostToDevice));
        FLS_HIP(hipMemcpy(dg.p, g, size_t(n) * 6 * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(debug_fullpiv_qr6_kernel, dim3(unsigned(n)), dim3(64), 0, nullptr, (const double*)dH.p, (const double*)dg.p, n, dx.p);
        FLS_HIP(hipGetLastError());
        FLS_HIP(hipMemcpy(x, dx.p, size_t(n) * 6 * sizeof(double), hipMemcpyDeviceToHost));
        return FLS_OK;
    });
}

fls_status fls_debug_ldlt6(int device_id, const double* H, const double* g, int n, double* x, int32_t* ok) {
    if (!H || !g || !x || !ok || n < 0) return FLS_ERR_INVALID;
    if (n == 0) return FLS_OK;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(device_id));
        DevBuf<double> dH, dg, dx;
        DevBuf<int> dok;
        dH.reserve(size_t(n) * 36); dg.reserve(size_t(n) * 6); dx.reserve(size_t(n) * 6); dok.reserve(size_t(n));
        FLS_HIP(hipMemcpy(dH.p, H, size_t(n) * 36 * sizeof(double), hipMemcpyHostToDevice));
        FLS_HIP(hipMemcpy(dg.p, g, size_t(n) * 6 * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(debug_ldlt6_kernel, dim3(unsigned(n)), dim3(64), 0, nullptr, (const double*)dH.p, (const double*)dg.p, n, dx.p, dok.p);
        FLS_HIP(hipGetLastError());
        FLS_HIP(hipMemcpy(x, dx.p, size_t(n) * 6 * sizeof(double), hipMemcpyDeviceToHost));
        FLS_HIP(hipMemcpy(ok, dok.p, size_t(n) * sizeof(int), hipMemcpyDeviceToHost));
        return FLS_OK;
    });
}

fls_status fls_debug_voxel_grid(int device_id, const float* pts, size_t n, int stride, float leaf, float* out, size_t cap, size_t* n_out) {
    if (!pts || !n_out || stride < 3 || !(leaf > 0.f) || (cap && !out)) return FLS_ERR_INVALID;
    *n_out = 0;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(device_id));
        DevScan raw;
        DeviceVoxelGrid vg;
        raw.upload_raw(pts, n, stride, nullptr, true);
        if (n == 0 || !vg.run(raw.x.p, raw.y.p, raw.z.p, raw.xyz.p + 3 * n, n, leaf, nullptr)) return FLS_ERR_STATE;
        *n_out = vg.n_out;
        if (vg.n_out > cap) return FLS_ERR_INVALID;
        std::vector<float> tmp;
        const std::vector<PtI> c = vg.download(nullptr, tmp);
        std::memcpy(out, c.data(), c.size() * sizeof(PtI));
        return FLS_OK;
    });
}

fls_status fls_debug_voxel_grid_timed(int device_id, const float* pts, size_t n, int stride, float leaf, int reps, double* ms, size_t* n_out) {
    if (!pts || !n_out || !ms || reps <= 0 || stride < 3 || !(leaf > 0.f)) return FLS_ERR_INVALID;
    *n_out = 0;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(device_id));
        DevScan raw;
        DeviceVoxelGrid vg;  // one filter object for all repetitions: the buffers are allocated by the first one, as in a matcher
        raw.upload_raw(pts, n, stride, nullptr, true);
        FLS_HIP(hipDeviceSynchronize());
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            const bool ok = n != 0 && vg.run(raw.x.p, raw.y.p, raw.z.p, raw.xyz.p + 3 * n, n, leaf, nullptr);
            FLS_HIP(hipStreamSynchronize(nullptr));
            ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (!ok) return FLS_ERR_STATE;
        }
        *n_out = vg.n_out;
        return FLS_OK;
    });
}

fls_status fls_debug_exact_sort(int device_id, uint32_t* key, uint32_t* val, size_t n, int on_host) {
    if ((!key || !val) && n) return FLS_ERR_INVALID;
    if (on_host != 0 && on_host != 1 && on_host != 2) return FLS_ERR_INVALID;
    if (on_host == 1)
        return guarded([&]() -> fls_status {  // (an allocation failure must not cross the C ABI: ADVICE r4)
            std::vector<VoxelLeafRec> r(n);
            for (size_t i = 0; i < n; ++i) r[i] = VoxelLeafRec{key[i], val[i]};
            std::sort(r.begin(), r.end());  // operator< compares idx only: libstdc++'s introsort decides the order of equal keys
            for (size_t i = 0; i < n; ++i) { key[i] = r[i].idx; val[i] = r[i].pt; }
            return FLS_OK;
        });
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(device_id));
        if (n == 0) return FLS_OK;
        DevBuf<unsigned> dk, dv;
        dk.reserve(n); dv.reserve(n);
        FLS_HIP(hipMemcpy(dk.p, key, n * sizeof(unsigned), hipMemcpyHostToDevice));
        FLS_HIP(hipMemcpy(dv.p, val, n * sizeof(unsigned), hipMemcpyHostToDevice));
        DeviceExactSort es;
        if (on_host == 2) {
            // the launch sequence the one-stream VoxelGrid queues (DeviceExactSort::fused_launch: pre-enqueued top levels + the task kernel, no host
            // round trip), with the initialisation vg_minmax_plan's last block does written from here
            if (n < 2) return FLS_OK;
            if (!es.fused_ok(n)) return FLS_ERR_STATE;
            const EsInitArgs ia = es.fused_prepare(n);
            FLS_HIP(hipMemset(ia.st, 0, sizeof(EsState)));
            FLS_HIP(hipMemset(ia.ready, 0, ia.work_cap * sizeof(unsigned)));
            const EsQueue q0{0u, 0u, 1u, 0u};
            FLS_HIP(hipMemcpy(ia.q, &q0, sizeof(EsQueue), hipMemcpyHostToDevice));
            es.fused_launch(dk.p, dv.p, n, nullptr, nullptr);
            FLS_HIP(hipDeviceSynchronize());
            EsState hs;
            FLS_HIP(hipMemcpy(&hs, ia.st, sizeof(EsState), hipMemcpyDeviceToHost));
            if (hs.fail != 0u) return FLS_ERR_STATE;
            FLS_HIP(hipMemcpy(key, dk.p, n * sizeof(unsigned), hipMemcpyDeviceToHost));
            FLS_HIP(hipMemcpy(val, dv.p, n * sizeof(unsigned), hipMemcpyDeviceToHost));
            return FLS_OK;
        }
        const bool queued = es.run(dk.p, dv.p, n, nullptr);
        FLS_HIP(hipDeviceSynchronize());
        if (!queued || es.failed_after_sync()) return FLS_ERR_STATE;
        FLS_HIP(hipMemcpy(key, dk.p, n * sizeof(unsigned), hipMemcpyDeviceToHost));
        FLS_HIP(hipMemcpy(val, dv.p, n * sizeof(unsigned), hipMemcpyDeviceToHost));
        return FLS_OK;
    });
}

extern "C" int fls_debug_exact_sort_marks(unsigned* out, int n) {  // diagnostics only (declared in fls_reg.h): progress stamps of the sort in flight
    if (!out || n <= 0) return -1;
    EsMailbox* m = es_debug_mailbox().load(std::memory_order_acquire);
    if (!m) return -1;
    for (int i = 0; i < n && i < 12; ++i) out[i] = __atomic_load_n(&m->mark[i], __ATOMIC_RELAXED);
    return 0;
}

fls_status fls_voxel_grid_cloud(int device_id, fls_voxelgrid_mode mode, const float* pts, size_t n, int stride, float leaf, float* out, size_t cap,
                                size_t* n_out) {
    if (mode == FLS_VOXELGRID_DEVICE) return fls_debug_voxel_grid(device_id, pts, n, stride, leaf, out, cap, n_out);
    if (mode != FLS_VOXELGRID_EXACT || (!pts && n) || !n_out || stride < 3 || !(leaf > 0.f) || (cap && !out)) return FLS_ERR_INVALID;
    *n_out = 0;
    return guarded([&]() -> fls_status {
        const std::vector<PtI> c = voxel_grid_strided(pts, n, stride, leaf);  // host_maps.hpp: the exact filter (pooled introsort order)
        *n_out = c.size();
        if (c.size() > cap) return FLS_ERR_INVALID;
        if (!c.empty()) std::memcpy(out, c.data(), c.size() * sizeof(PtI));
        return FLS_OK;
    });
}

fls_status fls_loop_match(int device_id, const float* source, size_t n_source, const float* target, size_t n_target, int stride, double T[16], float* fitness,
                          fls_loop_stats* stats) {
    if (!T || !fitness || stride < 3 || (!source && n_source) || (!target && n_target)) return FLS_ERR_INVALID;
    *fitness = std::numeric_limits<float>::max();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return guarded([&]() -> fls_status {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return FLS_ERR_DEVICE;
        hipDeviceProp_t prop;
        FLS_HIP(hipGetDeviceProperties(&prop, device_id));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FLS_ERR_DEVICE;
        const bool timing = std::getenv("FLS_HOST_TIMING") && std::atoi(std::getenv("FLS_HOST_TIMING")) != 0;
        const auto t0 = std::chrono::steady_clock::now();
        // one matcher per device, kept for the life of the process (stream, result block, device buffers: ~3 ms to set up, the
        // reference's loop-closure thread calls Match once per candidate); calls on one device are serialised (run_mx).  Never destroyed:
        // static destruction would run after the HIP runtime's own teardown.
        static std::mutex mx;
        static std::map<int, LoopMatcher*>* cache = new std::map<int, LoopMatcher*>();
        LoopMatcher* m = nullptr;
        {
            std::lock_guard<std::mutex> lk(mx);
            LoopMatcher*& slot = (*cache)[device_id];
            if (!slot) {
                std::unique_ptr<LoopMatcher> fresh(new LoopMatcher());
                fresh->init(device_id);
                slot = fresh.release();
            }
            m = slot;
        }
        std::lock_guard<std::mutex> run_lk(m->run_mx);  // (devices run side by side, calls on one device one after the other)
        if (timing) std::fprintf(stderr, "[fls loop] ms: matcher set-up %.2f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        const fls_status rc = m->run(cloud_from(source, n_source, stride), cloud_from(target, n_target, stride), T, fitness);
        if (stats) *stats = m->st;
        return rc;
    });
}

fls_status fls_get_traffic_counters(fls_handle h, uint64_t* probes, uint64_t* hit_voxels, uint64_t* cand_points) {
    if (!h) return FLS_ERR_INVALID;
    if (probes) *probes = h->last_tc.probes;
    if (hit_voxels) *hit_voxels = h->last_tc.hits;
    if (cand_points) *cand_points = h->last_tc.cand;
    return FLS_OK;
}

// ---- LOAM feature front-end (include/fls_features.h) ------------------------------------------------------------
fls_status fls_features_create(const fls_feature_params* params, int device_id, fls_features_handle* out) {
    if (!out) return FLS_ERR_INVALID;
    *out = nullptr;
    if (!params) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return FLS_ERR_DEVICE;
        hipDeviceProp_t prop;
        FLS_HIP(hipGetDeviceProperties(&prop, device_id));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FLS_ERR_DEVICE;
        std::unique_ptr<fls_features> f(new fls_features());
        f->p = *params;
        f->device = device_id;
        const fls_status rc = f->init();
        if (rc != FLS_OK) return rc;
        *out = f.release();
        return FLS_OK;
    });
}

void fls_features_destroy(fls_features_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    delete h;
}

fls_status fls_features_project(fls_features_handle h, const void* raw, size_t n, const fls_point_layout* layout, size_t* n_ordered) {
    if (!h || !layout || (!raw && n)) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->project(raw, n, *layout, n_ordered);
    });
}

fls_status fls_features_extract(fls_features_handle h, size_t* n_corner, size_t* n_planar) {
    if (!h) return FLS_ERR_INVALID;
    return guarded([&]() -> fls_status {
        FLS_HIP(hipSetDevice(h->device));
        return h->extract(n_corner, n_planar);
    });
}

size_t fls_features_get(fls_features_handle h, int what, void* out, size_t cap_elems) {
    if (!h) return 0;
    size_t n = 0;
    (void)guarded([&]() -> fls_status { n = h->get(what, out, cap_elems); return FLS_OK; });  // may read an introspection array back
    return n;
}

fls_status fls_features_get_time(fls_features_handle h, double* project_ms, double* extract_ms) {
    if (!h) return FLS_ERR_INVALID;
    if (project_ms) *project_ms = h->project_ms;
    if (extract_ms) *extract_ms = h->extract_ms;
    return FLS_OK;
}

}  // extern "C"
