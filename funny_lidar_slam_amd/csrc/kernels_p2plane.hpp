// kernels_p2plane.hpp -- the point-to-plane correspondence + residual kernel against the
// iVox hash-voxel map, and the LOAM-family Gauss-Newton solve kernel.   gfx950 only.
//
// p2plane_ivox_kernel replaces, per Gauss-Newton iteration:
//   LoamPointToPlaneIVOX::PlanerMatch           loam_point_to_plane_ivox.h:256-324
//     pcl::transformPoint(double -> float)       :265-266
//     IVoxMap::GetClosestPoint (19-voxel kNN-5)  src/ivox_map/ivox_map.cpp:6-37
//       IVoxMap::Pos2Grid                        src/ivox_map/ivox_map.cpp:145-147
//       VoxelGridNode::KNNPointByCondition       src/ivox_map/voxel_grid_node.cpp:23-42
//       DistanceSquared (float)                  include/common/pointcloud_utility.h:14-17
//     5x3 plane fit + gates + Jacobian           :275-321
//   and the per-wave part of SumCoefficient      :326-340
// gn_solve_loam_kernel finishes SumCoefficient (fixed-order reduction of the wave partials) and runs
//   dx = H.fullPivHouseholderQr().solve(g); R <- Exp(dx[0:3]) R; t += dx[3:6]; stop rule   :167-195
// on the device so that a whole Match needs one host synchronisation.
//
// One lane = one source point.  One 64-lane workgroup = one wave: the 6x6 normal equations are
// reduced with wavefront shuffles only (no LDS, no barrier, no atomics -> bit-reproducible).
// No MFMA: there is no dense contraction in this path (21+6+2 scalars per point).
#pragma once
#include "device_common.hpp"
#include "linalg_dev.hpp"

namespace fls {

// IVoxMap::GenerateNearbyGrids NEARBY18 order (ivox_map.cpp:50-54)
__device__ constexpr signed char kNearby18[19][3] = {
    {0, 0, 0},  {-1, 0, 0}, {1, 0, 0},  {0, 1, 0},  {0, -1, 0}, {0, 0, -1}, {0, 0, 1},  {1, 1, 0},  {-1, 1, 0}, {1, -1, 0},
    {-1, -1, 0}, {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1}, {0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}};

// Point-to-plane residual on 5 neighbours (Appendix C.1).  nn[0] must be the nearest neighbour.
// Returns false wherever the reference lambda returns early.
__device__ __forceinline__ bool plane_residual_dev(const float4 (&nn)[5], const float spx, const float spy, const float spz,
                                                   const float ptx, const float pty, const float ptz, const double* __restrict__ T,
                                                   const double thres, double (&J)[6], double& d_abs) {
    double A[3][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { A[0][j] = (double)nn[j].x; A[1][j] = (double)nn[j].y; A[2][j] = (double)nn[j].z; }
    double x[3];
    plane_fit_5x3(A, x);
    const double nrm = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double r = ((A[0][j] * x[0] + A[1][j] * x[1]) + A[2][j] * x[2]) + 1.0;
        if (fabs(r) / nrm > thres) ok = false;
    }
    if (!ok) return false;
    const double n0 = x[0] / nrm, n1 = x[1] / nrm, n2 = x[2] / nrm;
    const double ps0 = (double)spx, ps1 = (double)spy, ps2 = (double)spz;
    const double d = (((double)ptx - A[0][0]) * n0 + ((double)pty - A[1][0]) * n1) + ((double)ptz - A[2][0]) * n2;
    const double range = sqrt((ps0 * ps0 + ps1 * ps1) + ps2 * ps2);
    if (range < 81 * d * d) return false;
    const double s = d > 0 ? 1.0 : -1.0;
    const double v0 = (T[0] * ps0 + T[4] * ps1) + T[8] * ps2;
    const double v1 = (T[1] * ps0 + T[5] * ps1) + T[9] * ps2;
    const double v2 = (T[2] * ps0 + T[6] * ps1) + T[10] * ps2;
    J[0] = ((0.0 * n0 + (-v2) * n1) + v1 * n2) * s;
    J[1] = ((v2 * n0 + 0.0 * n1) + (-v0) * n2) * s;
    J[2] = (((-v1) * n0 + v0 * n1) + 0.0 * n2) * s;
    J[3] = n0 * s;
    J[4] = n1 * s;
    J[5] = n2 * s;
    d_abs = fabs(d);
    return true;
}

// wave-reduce the rank-1 contribution (J J^T, -J r, r, 1) of every lane and store one partial row
__device__ __forceinline__ void reduce_rank1_and_store(const bool contrib, const double (&J)[6], const double r,
                                                       double* __restrict__ partial_row) {
    const int lane = threadIdx.x & 63;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            const double v = wave_sum(contrib ? J[a] * J[b] : 0.0);
            if (lane == 0) partial_row[k] = v;
            ++k;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double v = wave_sum(contrib ? (-J[a]) * r : 0.0);
        if (lane == 0) partial_row[21 + a] = v;
    }
    const double sr = wave_sum(contrib ? r : 0.0);
    const double sc = wave_sum(contrib ? 1.0 : 0.0);
    if (lane == 0) { partial_row[27] = sr; partial_row[28] = sc; }
}

template <bool COUNT>
__global__ void __launch_bounds__(64)
p2plane_ivox_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                    const GnState* __restrict__ st, const DevGrid grid, const float inv_res,
                    float4* __restrict__ nn_pts /* [n][5] */, unsigned char* __restrict__ nn_cnt,
                    double* __restrict__ Jst /* [7][n] */, unsigned char* __restrict__ flag,
                    double* __restrict__ partials, const double plane_thres, TrafficCounters* __restrict__ tc) {
    if (st->done) return;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool active = i < n;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->T[(k / 3) * 4 + (k % 3)];  // T[c*3+r] -> remapped below
    // local compact copy: Tc[r + 4*c] access pattern used by plane_residual_dev wants the 4x4 layout
    double T44[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) T44[r + 4 * c] = T[c * 3 + r];

    bool contrib = false;
    double J[6] = {0, 0, 0, 0, 0, 0};
    double res = 0.0;
    unsigned long long c_probes = 0, c_hits = 0, c_cand = 0;

    if (active) {
        const float px = sx[i], py = sy[i], pz = sz[i];
        const double x = px, y = py, z = pz;
        // pcl::transformPoint: evaluate in double left-to-right, round to float
        const float ptx = (float)(((T44[0] * x + T44[4] * y) + T44[8] * z) + T44[12]);
        const float pty = (float)(((T44[1] * x + T44[5] * y) + T44[9] * z) + T44[13]);
        const float ptz = (float)(((T44[2] * x + T44[6] * y) + T44[10] * z) + T44[14]);
        // Pos2Grid: float multiply, round half away from zero, to int
        const float fx = roundf(ptx * inv_res), fy = roundf(pty * inv_res), fz = roundf(ptz * inv_res);
        const bool in_range = fabsf(fx) < (float)kKeyLimit && fabsf(fy) < (float)kKeyLimit && fabsf(fz) < (float)kKeyLimit;
        const int kx = in_range ? (int)fx : 0, ky = in_range ? (int)fy : 0, kz = in_range ? (int)fz : 0;

        float bd[5] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
        unsigned bs[5] = {0, 0, 0, 0, 0};
        int ncand = 0;
        if (in_range) {
            // phase 1: issue all 19 first-slot probes (independent loads in flight together)
            HashEntry e[19];
            unsigned hh[19];
#pragma unroll
            for (int k = 0; k < 19; ++k) {
                const unsigned long long key = pack_key(kx + kNearby18[k][0], ky + kNearby18[k][1], kz + kNearby18[k][2]);
                hh[k] = hash_key(key) & grid.mask;
                e[k] = grid.table[hh[k]];
            }
            // phase 2: resolve (linear probing on collision) and scan the voxel's points
#pragma unroll
            for (int k = 0; k < 19; ++k) {
                const unsigned long long key = pack_key(kx + kNearby18[k][0], ky + kNearby18[k][1], kz + kNearby18[k][2]);
                HashEntry ek = e[k];
                unsigned h = hh[k];
                while (ek.key != key && ek.key != kEmptyKey) {
                    h = (h + 1) & grid.mask;
                    ek = grid.table[h];
                }
                if (COUNT) c_probes++;
                if (ek.key == key) {
                    if (COUNT) { c_hits++; c_cand += ek.count; }
                    const unsigned end = ek.begin + ek.count;
                    for (unsigned s = ek.begin; s < end; ++s) {
                        const float4 q = grid.pts[s];
                        const float dx = q.x - ptx, dy = q.y - pty, dz = q.z - ptz;
                        const float d2 = dx * dx + (dy * dy + dz * dz);  // Eigen Vector3f::squaredNorm order
                        if (d2 < 25.0f) {  // max_range 5.0 squared (never binds at 0.5 m voxels)
                            ++ncand;
                            top5_insert(bd, bs, d2, s);
                        }
                    }
                }
            }
        }
        int cnt;
        float4 nn[5];
        if (ncand > 0) {
            cnt = ncand < 5 ? ncand : 5;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                nn[j] = (j < cnt) ? grid.pts[bs[j]] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
                nn_pts[(size_t)i * 5 + j] = nn[j];
            }
            nn_cnt[i] = (unsigned char)cnt;
        } else {
            // ivox_map.cpp:21-23: "return false" BEFORE closest_pt.clear(): the previous list survives
            cnt = nn_cnt[i];
            if (cnt == 5) {
#pragma unroll
                for (int j = 0; j < 5; ++j) nn[j] = nn_pts[(size_t)i * 5 + j];
            }
        }
        bool valid_now = false;
        if (cnt == 5) valid_now = plane_residual_dev(nn, px, py, pz, ptx, pty, ptz, T44, plane_thres, J, res);
        if (valid_now) {
            // Q1: per-point slots persist across iterations (flags are cleared once per Match)
#pragma unroll
            for (int a = 0; a < 6; ++a) Jst[(size_t)a * n + i] = J[a];
            Jst[(size_t)6 * n + i] = res;
            flag[i] = 1;
            contrib = true;
        } else if (flag[i]) {
#pragma unroll
            for (int a = 0; a < 6; ++a) J[a] = Jst[(size_t)a * n + i];
            res = Jst[(size_t)6 * n + i];
            contrib = true;
        }
    }
    reduce_rank1_and_store(contrib, J, res, partials + (size_t)blockIdx.x * kPartialStride);
    if (COUNT) {
        const double p = wave_sum((double)c_probes), h = wave_sum((double)c_hits), c = wave_sum((double)c_cand);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&tc->probes, (unsigned long long)p);
            atomicAdd(&tc->hits, (unsigned long long)h);
            atomicAdd(&tc->cand, (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fixed-order reduction of wave partial rows: rows [0,nrows) of `partials`, 32 columns.
// blockDim = 1024 = 32 row-groups x 32 columns.  Result in tot[32] (LDS), valid after the call.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, const int nrows, double* tot /*LDS 32*/,
                                                double (*red)[33] /*LDS 32x33*/) {
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double s = 0.0;
    for (int r = grp; r < nrows; r += 32) s += partials[(size_t)r * kPartialStride + col];
    red[grp][col] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int g = 0; g < 32; ++g) t += red[g][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// LOAM family tail: loam_point_to_plane_ivox.h:160-196 (same in loam_full_kdtree.h:121-176,
// loam_point_to_plane_kdtree.h:99-136).  partials_a (corner rows, may be empty) are summed before
// partials_b (planar rows), as SumCoefficient does (loam_full_kdtree.h:347-372).
__global__ void __launch_bounds__(1024)
gn_solve_loam_kernel(GnState* __restrict__ st, const double* __restrict__ partials_a, const int nrows_a,
                     const double* __restrict__ partials_b, const int nrows_b, const double rot_thr, const double pos_thr) {
    if (st->done) return;
    __shared__ double red[32][33];
    __shared__ double tot_a[32], tot_b[32];
    __shared__ double Hs[36], gs[6], xs[6], hc[6];
    __shared__ int tr[6], perm[6];
    if (nrows_a > 0) reduce_partials(partials_a, nrows_a, tot_a, red);
    else { if (threadIdx.x < 32) tot_a[threadIdx.x] = 0.0; __syncthreads(); }
    reduce_partials(partials_b, nrows_b, tot_b, red);
    if (threadIdx.x == 0) {
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) {
                const double v = tot_a[k] + tot_b[k];
                Hs[a + b * 6] = v;
                Hs[b + a * 6] = v;
                ++k;
            }
        for (int a = 0; a < 6; ++a) gs[a] = tot_a[21 + a] + tot_b[21 + a];
        for (int q = 0; q < 36; ++q) st->H[q] = Hs[q];
        for (int q = 0; q < 6; ++q) st->g[q] = gs[q];
        fullpiv_qr_solve6(Hs, gs, xs, hc, tr, perm);
        double Rd[9], R[9], Rn[9];
        so3_exp_dev(xs, Rd);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) R[i + j * 3] = st->T[i + j * 4];
        mat3_mul_dev(Rd, R, Rn);  // left-multiplicative
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) st->T[i + j * 4] = Rn[i + j * 3];
        st->T[12] += xs[3];
        st->T[13] += xs[4];
        st->T[14] += xs[5];
        const double rn = norm3d(xs), pn = norm3d(xs + 3);
        const double drot = fabs(rn - st->last_rot), dpos = fabs(pn - st->last_pos);
        st->last_rot = rn;
        st->last_pos = pn;
        for (int q = 0; q < 6; ++q) st->last_dx[q] = xs[q];
        const int it = st->iter;
        st->n_valid = (int)tot_b[28];
        st->n_valid2 = (int)tot_a[28];
        st->sum_res = tot_b[27];
        st->sum_res2 = tot_a[27];
        if (it < kMaxIter) {
            for (int q = 0; q < 16; ++q) st->log_T[it][q] = st->T[q];
            st->log_nv[it] = (int)tot_b[28];
            st->log_res[it] = tot_b[27];
        }
        st->iter = it + 1;
        if ((rn < rot_thr && pn < pos_thr) || (drot < 1.0e-4 && dpos < 1.0e-4)) st->done = 1;
    }
}

}  // namespace fls
