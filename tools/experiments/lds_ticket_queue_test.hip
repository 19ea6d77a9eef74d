// standalone check of the LDS ticket-queue protocol used by es_phase_b (kernels_exactsort.hpp): hipcc --offload-arch=gfx950 -O3 -o /tmp/tq lds_ticket_queue_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kQ = 1024;
__global__ void __launch_bounds__(1024) tq(unsigned* out, int n_init, int fan) {
    __shared__ unsigned qt[kQ];
    __shared__ unsigned head, tail, open, fail, work;
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < kQ; i += 1024) qt[i] = 0u;
    if (t == 0) { head = 0; tail = n_init; open = n_init; fail = 0; work = 0; }
    __syncthreads();
    if (t < n_init) qt[t] = 8u;  // a task = remaining depth
    __syncthreads();
    for (;;) {
        // (no divergent region may end at the loop's back edge: the structurizer let lane 0 and lanes 1..63 take it separately, and the
        // readfirstlane at the top then ran per group -- round 5; every lane executes every atomic, only lane 0 adds something)
        const unsigned t0 = atomicAdd(&head, lane == 0 ? 1u : 0u);
        const unsigned ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
        unsigned word = 0u;
        if (ticket < (unsigned)kQ) {
            for (unsigned spin = 0;; ++spin) {
                word = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&qt[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (word != 0u) break;
                if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) break;
                if (spin > 400000u) { __hip_atomic_store(&fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (word == 0u) break;
        // "work": spawn `fan` children of depth - 1
        atomicAdd(&work, lane == 0 ? 1u : 0u);
        if (word > 1u) {
            for (int c = 0; c < fan; ++c) {
                atomicAdd(&open, lane == 0 ? 1u : 0u);
                const unsigned s2 = (unsigned)__builtin_amdgcn_readfirstlane((int)atomicAdd(&tail, lane == 0 ? 1u : 0u));
                if (s2 < (unsigned)kQ) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __hip_atomic_store(&qt[s2], word - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                else atomicSub(&open, lane == 0 ? 1u : 0u);
            }
        }
        atomicSub(&open, lane == 0 ? 1u : 0u);
    }
    __syncthreads();
    if (t == 0) { out[0] = head; out[1] = tail; out[2] = open; out[3] = fail; out[4] = work; }
}
int main() {
    unsigned* d; hipMalloc(&d, 64); printf("init done\n"); fflush(stdout);
    for (int fan = 0; fan <= 2; ++fan) for (int n_init = 1; n_init <= 3; n_init += 2) {
        hipMemset(d, 0xff, 64);
        tq<<<1, 1024>>>(d, n_init, fan);
        hipError_t e = hipDeviceSynchronize();
        unsigned h[5]; hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
        printf("fan %d init %d: err %d head %u tail %u open %u fail %u work(lane-adds) %u\n", fan, n_init, (int)e, h[0], h[1], h[2], h[3], h[4]); fflush(stdout);
    }
    return 0;
}
