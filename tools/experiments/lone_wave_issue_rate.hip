// lone_wave_issue_rate.hip -- how fast does ONE wave issue on gfx950?  (round 6: the Gauss-Newton tails are one wave alone on its SIMD.)
// Chains of N FP64 / FP32 FMAs, dependent (1 chain) or interleaved (2, 4, 8 independent chains), timed with the shader clock
// (s_memtime), with 1, 2 or 4 waves per SIMD resident (workgroups of 256 / 512 / 1024 threads on one CU).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/lone_wave_issue_rate tools/experiments/lone_wave_issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double fma_t(double x, double a, double b) { return __builtin_fma(x, a, b); }
__device__ __forceinline__ float fma_t(float x, float a, float b) { return __builtin_fmaf(x, a, b); }  // (the first run called __builtin_fma on floats: an f64 FMA between two conversions, 3 instructions -- its f32 lines are that)
template <typename T, int CH>
__global__ void chain_kernel(T* out, long long* ticks, const int n, const T a, const T b) {
    T x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = (T)(threadIdx.x + c);
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fma_t(x[c], a, b);
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    T s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <typename T, int CH>
static void run(const char* name, const int threads) {
    T* out; long long* ticks;
    hipMalloc(&out, sizeof(T) * threads); hipMalloc(&ticks, sizeof(long long) * 16);
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((chain_kernel<T, CH>), dim3(1), dim3(threads), 0, nullptr, out, ticks, n, (T)0.999, (T)0.001);
    hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    hipMemcpy(h.data(), ticks, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
    long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    std::printf("%-4s chains %d  waves/SIMD %d : %.2f ticks per FMA per wave, %.2f ticks per FMA per SIMD\n", name, CH, threads / 256, (double)mx / ((double)n * CH),
                (double)mx / ((double)n * CH * (threads / 256)));
    hipFree(out); hipFree(ticks);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        run<double, 1>("f64", threads); run<double, 2>("f64", threads); run<double, 4>("f64", threads); run<double, 8>("f64", threads);
        run<float, 1>("f32", threads); run<float, 2>("f32", threads); run<float, 4>("f32", threads); run<float, 8>("f32", threads);
    }
    return 0;
}
