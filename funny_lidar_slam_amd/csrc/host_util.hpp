// host_util.hpp -- host-side plumbing: HIP error handling, device buffers, pinned staging.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <stdexcept>
#include "../../include/fls_reg.h"

namespace fls {

struct HipError : std::runtime_error {
    hipError_t code;
    HipError(hipError_t c, const char* what) : std::runtime_error(what), code(c) {}
};

#define FLS_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            char _buf[256];                                                                    \
            snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw ::fls::HipError(_e, _buf);                                                   \
        }                                                                                      \
    } while (0)

// growable device buffer; contents are NOT preserved on growth unless keep=true
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void reserve(size_t n, bool keep = false, hipStream_t s = nullptr, bool zero_new = false) {
        if (n <= cap) return;
        size_t nc = cap ? cap : 1;
        while (nc < n) nc = nc + nc / 2 + 64;
        T* q = nullptr;
        FLS_HIP(hipMalloc(&q, nc * sizeof(T)));
        if (zero_new) FLS_HIP(hipMemsetAsync(q, 0, nc * sizeof(T), s));
        if (keep && p && cap) FLS_HIP(hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s));
        // growth is rare (buffers only grow): wait for the WHOLE device before the old allocation goes away -- kernels still
        // queued on any of the handle's streams (e.g. exit-at-once launches behind a converged Match) may hold the pointer
        if (p) { FLS_HIP(hipDeviceSynchronize()); FLS_HIP(hipFree(p)); }
        p = q;
        cap = nc;
    }
};

template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) FLS_HIP(hipHostFree(p));
        p = nullptr;
        size_t nc = n + n / 4 + 64;
        FLS_HIP(hipHostMalloc(&p, nc * sizeof(T), hipHostMallocDefault));
        cap = nc;
    }
};

}  // namespace fls
