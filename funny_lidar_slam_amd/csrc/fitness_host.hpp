// fitness_host.hpp -- GetFitnessScore shared by the kinds that support it
// (icp_optimized.h:191-215, loam_point_to_plane_ivox.h:225-253, loam_point_to_plane_kdtree.h:160-183,
//  incremental_ndt.h:345-372): float-transform the last source with the final pose, 1-NN squared
// distance on the device, then the reference's SEQUENTIAL float accumulation on the host so the
// score is bit-identical to an index-order loop.
#pragma once
#include "matcher_base.hpp"
#include "kernels_knn.hpp"

namespace fls {

inline CellGridDev cell_dev(const CellGridImage& g) { return CellGridDev{g.dev(), 1.0 / double(g.cell), double(g.cell), g.rings, g.window(), g.d_by_id.p}; }

inline fls_status fitness_score_device(fls_matcher& m, const CellGridImage& grid, const DevScan& scan, const double* final_T,
                                       float max_range, float* score) {
    const size_t n = scan.n;
    if (n == 0) { *score = std::numeric_limits<float>::max(); return FLS_OK; }
    DevBuf<float> d_d2;
    DevBuf<double> d_T;
    d_d2.reserve(n);
    d_T.reserve(16);
    FLS_HIP(hipMemcpyAsync(d_T.p, final_T, 16 * sizeof(double), hipMemcpyHostToDevice, m.stream));
    const int nblk = int((n + 63) / 64);
    hipLaunchKernelGGL(nn_dist_kernel, dim3(nblk), dim3(64), 0, m.stream, scan.x.p, scan.y.p, scan.z.p, int(n), d_T.p, cell_dev(grid),
                       max_range, d_d2.p);
    FLS_HIP(hipGetLastError());
    std::vector<float> d2(n);
    FLS_HIP(hipMemcpyAsync(d2.data(), d_d2.p, n * sizeof(float), hipMemcpyDeviceToHost, m.stream));
    FLS_HIP(hipStreamSynchronize(m.stream));
    float s = 0.0f;
    int nr = 0;
    for (size_t i = 0; i < n; ++i)
        if (d2[i] <= max_range) { s += d2[i]; nr++; }
    *score = nr > 0 ? s / float(nr) : std::numeric_limits<float>::max();
    return FLS_OK;
}

}  // namespace fls
