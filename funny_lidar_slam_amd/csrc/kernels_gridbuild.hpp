// kernels_gridbuild.hpp -- the exact-kNN cell grid of the kd-tree kinds (the stand-in for pcl::KdTreeFLANN::setInputCloud at
// icp_optimized.h:188, loam_full_kdtree.h:102-103, loam_point_to_plane_kdtree.h:78) built ON THE DEVICE.  SURVEY.md 8f rank 2.
//
// The image the search kernels read is {cells[nz][ny][nx] = {begin, count}, pts[n] = {x, y, z, map index}} with the points of a
// cell contiguous and in ascending map index.  That is a sort of (cell, index) pairs -- a TOTAL order, so any sort gives the same
// buckets; the host build (host_maps.hpp CellGridImage::build) does it with std::sort + a bucket loop + two uploads, this file does
// it with the stable LSD radix sort of kernels_voxelgrid.hpp on the linear cell index.  The buckets sit in (z, y, x) cell order
// here and in packed-key order on the host path: the ORDER OF THE BUCKETS in the point array is the one thing that differs, and
// nothing reads it -- every selection key of the search kernels carries the MAP INDEX as its tie-breaker, not the slot.
//
//   cg_keys     cell of every point (floorf(p * inv_cell) in float, like the host build) -> linear window index, value = point index
//   (vg_hist / vg_scan_rows / vg_scatter: ceil(bits / 8) radix passes)
//   cg_fill     points gathered into sorted order; the head of every run writes its cell's `begin`
//   cg_count    the tail of every run writes its cell's `count`
//   cg_by_id    optional: the cloud in its own order as float4 (grid_knn27_kernel gathers its winners there)
//   soa_append  AoS {x, y, z, i} rows -> four SoA planes at an offset (the device-side cloud deque)
#pragma once
#include "kernels_voxelgrid.hpp"

namespace fls {

struct CgWindow {
    float inv_cell;
    int ox, oy, oz, nx, ny, nz;
};

__global__ void __launch_bounds__(kVgBlock)
cg_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, const CgWindow w,
        unsigned* __restrict__ key, unsigned* __restrict__ val) {
    const int i = blockIdx.x * kVgBlock + threadIdx.x;
    if (i >= n) return;
    // the host decided the window from the finite bounds of these very points: every cell index is inside it
    const int cx = (int)floorf(x[i] * w.inv_cell) - w.ox, cy = (int)floorf(y[i] * w.inv_cell) - w.oy, cz = (int)floorf(z[i] * w.inv_cell) - w.oz;
    key[i] = (unsigned)((cz * w.ny + cy) * w.nx + cx);
    val[i] = (unsigned)i;
}

__global__ void __launch_bounds__(kVgBlock)
cg_fill(const unsigned* __restrict__ key, const unsigned* __restrict__ val, const int n, const float* __restrict__ x, const float* __restrict__ y,
        const float* __restrict__ z, float4* __restrict__ pts, uint2* __restrict__ cells) {
    const int a = blockIdx.x * kVgBlock + threadIdx.x;
    if (a >= n) return;
    const unsigned k = key[a], v = val[a];
    pts[a] = make_float4(x[v], y[v], z[v], __int_as_float((int)v));
    if (a == 0 || key[a - 1] != k) cells[k].x = (unsigned)a;
}

__global__ void __launch_bounds__(kVgBlock)
cg_count(const unsigned* __restrict__ key, const int n, uint2* __restrict__ cells) {
    const int a = blockIdx.x * kVgBlock + threadIdx.x;
    if (a >= n) return;
    const unsigned k = key[a];
    if (a == n - 1 || key[a + 1] != k) cells[k].y = (unsigned)(a + 1) - cells[k].x;
}

__global__ void __launch_bounds__(kVgBlock)
cg_by_id(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int n, float4* __restrict__ out) {
    const int i = blockIdx.x * kVgBlock + threadIdx.x;
    if (i < n) out[i] = make_float4(x[i], y[i], z[i], __int_as_float(i));
}

__global__ void __launch_bounds__(kVgBlock)
soa_append(const float4* __restrict__ rows, const int n, float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, float* __restrict__ in) {
    const int i = blockIdx.x * kVgBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = rows[i];
    x[i] = p.x; y[i] = p.y; z[i] = p.z; in[i] = p.w;
}

}  // namespace fls
