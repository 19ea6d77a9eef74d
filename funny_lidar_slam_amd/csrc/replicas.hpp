// replicas.hpp -- one process, several GPUs: the native host side of BASELINE configs[4] / SURVEY.md 8e ("one process + N host
// threads ... C++ host calling HIP through a thin C-ABI").  A replica set holds one handle per listed device, each carrying a READ-ONLY
// copy of the owner handle's map: since round 4 the owner's device image itself (points + brick directory + slabs: ~40 MB for the
// 1e6-point map), copied device to device -- hipMemcpyPeer over xGMI between GPUs -- with no export blob and no host mirror per replica
// (fls_matcher::replicate_from; round 3 exported a host blob once and imported it per device: 350 + 138 ms, still available as
// FLS_REPLICAS_VIA_BLOB=1 and still what separate processes do).  fls_replicas_match_batch block-partitions the jobs over the devices
// (the same partition as batch.py), runs fls_match_batch per device on its own host thread and writes every result straight into the
// caller's arrays: no collective, no gather step -- the ranks share an address space.
// torch.distributed (one process per GPU, bench.py --gpus N) remains the multi-process form of the same sharding.
#pragma once
#include "../../include/fls_reg.h"
#include "matcher_base.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

struct fls_replicas {
    // joins whatever was started, also when starting the next thread throws (std::system_error): a joinable std::thread must never be destroyed
    struct Threads {
        std::vector<std::thread> th;
        template <class F> void start(F&& f) { th.emplace_back(std::forward<F>(f)); }
        void join() { for (auto& t : th) if (t.joinable()) t.join(); }
        ~Threads() { join(); }
    };
    fls_handle owner = nullptr;
    std::vector<int> devices;
    std::vector<fls_handle> handles;  // handles[i] serves devices[i]; the owner itself serves the first entry that names its device
    std::vector<char> owned;          // created (and destroyed) by the set
    std::vector<double> import_ms;    // last replication, per entry (0 for the owner's own entry)

    ~fls_replicas() {
        for (size_t i = 0; i < handles.size(); ++i)
            if (owned[i] && handles[i]) fls_destroy(handles[i]);
    }

    // (re-)replicate the owner's current map.  Default: every replica copies the owner's DEVICE image, device to device (no export blob,
    // no host mirror per replica: a replica serves fls_match_batch only).  FLS_REPLICAS_VIA_BLOB=1 (A/B, round 3): fls_map_export once,
    // fls_map_import per device -- what separate processes do (batch.py), with full handles as the result.
    fls_status refresh() {
        if (!owner->can_replicate()) return FLS_ERR_STATE;  // kind without a replicable image
        bool any_owned = false;
        for (size_t i = 0; i < handles.size(); ++i) any_owned = any_owned || owned[i];
        if (!any_owned) return FLS_OK;  // the owner serves every entry itself
        bool via_blob = false;
        if (const char* e = std::getenv("FLS_REPLICAS_VIA_BLOB")) via_blob = std::atoi(e) != 0;
        if (!via_blob) {
            // one after the other: every copy reads the owner's HBM, and the owner's device is made current in between
            bool peer_copy_failed = false;
            for (size_t i = 0; i < handles.size() && !peer_copy_failed; ++i) {
                if (!owned[i]) continue;
                const auto t0 = std::chrono::steady_clock::now();
                try {
                    const fls_status rc = handles[i]->replicate_from(*owner);
                    if (rc < 0) return rc;
                } catch (const fls::HipError& e) {
                    // a platform that refuses the device-to-device copy (no peer path between two GPUs): the same image travels through host
                    // memory instead -- a different transport for the same bytes, said out loud
                    std::fprintf(stderr, "[fls_reg] replica set: device-to-device image copy %d -> %d failed (%s); using the host blob\n", owner->device,
                                 handles[i]->device, e.what());
                    (void)hipGetLastError();
                    peer_copy_failed = true;
                }
                import_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            if (!peer_copy_failed) return FLS_OK;
        }
        const size_t need = fls_map_export(owner, nullptr, 0);
        if (need == 0) return FLS_ERR_STATE;  // no exportable image
        std::vector<unsigned char> blob(need);
        if (fls_map_export(owner, blob.data(), blob.size()) != need) return FLS_ERR_STATE;
        std::vector<fls_status> rc(handles.size(), FLS_OK);
        Threads th;
        for (size_t i = 0; i < handles.size(); ++i) {
            if (!owned[i]) continue;
            th.start([&, i] {
                const auto t0 = std::chrono::steady_clock::now();
                rc[i] = fls_map_import(handles[i], blob.data(), blob.size());
                import_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            });
        }
        th.join();
        for (fls_status r : rc)
            if (r < 0) return r;
        return FLS_OK;
    }

    static void block(size_t n_jobs, size_t world, size_t rank, size_t& begin, size_t& end) {  // batch.py::partition
        const size_t base = n_jobs / world, extra = n_jobs % world;
        begin = rank * base + std::min(rank, extra);
        end = begin + base + (rank < extra ? 1 : 0);
    }

    fls_status match_batch(size_t n_jobs, const float* const* src0, const size_t* n0, const float* const* src1, const size_t* n1, int stride, double* T,
                           fls_stats* stats, int32_t* status, int lanes) {
        const size_t world = handles.size();
        std::vector<fls_status> rc(world, FLS_OK);
        Threads th;
        for (size_t r = 0; r < world; ++r) {
            size_t b, e;
            block(n_jobs, world, r, b, e);
            if (b == e) continue;
            th.start([&, r, b, e] {
                rc[r] = fls_match_batch(handles[r], e - b, src0 + b, n0 + b, src1 ? src1 + b : nullptr, n1 ? n1 + b : nullptr, stride, T + 16 * b, stats ? stats + b : nullptr,
                                        status ? status + b : nullptr, lanes);
            });
        }
        th.join();
        for (fls_status r : rc)
            if (r < 0) return r;
        return FLS_OK;
    }
};
