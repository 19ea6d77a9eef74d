// kernels_knn.hpp -- exact k-NN on a uniform hash grid (the GPU stand-in for the reference's
// pcl::KdTreeFLANN) and the residual kernels built on it.   gfx950 only.
//
// Exactness (SURVEY.md 0 note 3): cells are floor(double(p) * inv_cell) per axis.  After scanning
// every cell whose Chebyshev cell distance to the query cell is <= rho, every map point within
// Euclidean distance rho*cell of the query has been seen.  The search stops when
//   (a) K points are held and the K-th d2 <= (rho*cell)^2 (nothing unseen can beat it), or
//   (b) (rho*cell)^2 >= gate (the caller rejects anything farther than `gate`, a SQUARED distance:
//       icp_optimized.h:87, loam_full_kdtree.h:227,291, fitness loops), or
//   (c) rho == kMaxRing, after which the query falls back to a brute-force scan of the whole map
//       (only reachable for the un-gated LoamPointToPlaneKdtree search in nearly empty space).
// Distances are FLANN L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz; ties resolve to the lower map
// index (FLANN's order among equal distances is traversal-defined; the oracle fixes the same rule).
//
// This header: the serial ring search (fallback of the cooperative kernel in kernels_grid_coop.hpp, and the
// fitness loops), the LOAM point-to-line residual, and
//   ndt_kernel            IncrementalNDT::Match per-point lambda          incremental_ndt.h:252-285
//   nn_dist_kernel        GetFitnessScore loops                           icp_optimized.h:191-215 etc.
//   gn_solve_lu_kernel    H.inverse()*B + right-multiplicative update     icp_optimized.h:129-148, incremental_ndt.h:306-322
#pragma once
#include "kernels_p2plane.hpp"

namespace fls {

constexpr int kMaxRing = 4;

struct CellGridDev {
    DevGrid g;
    double inv_cell;
    double cell;
    int rings;  // 1: the gate fits into one cell (27-cell block suffices); 2: half-size cells, 125-cell block in two stages
    DenseWindow win;  // cells == nullptr: no dense window (extent too large), hash table only
    const float4* by_id;  // the map cloud in its own order {x, y, z, id bits} (grid_knn27_kernel gathers its winners here), may be null
};

template <int K>
struct KnnResult {
    float d[K];
    int id[K];
    unsigned slot[K];
    int found;  // number of valid entries (<= K)
};

template <int K>
__device__ __forceinline__ void knn_insert(KnnResult<K>& r, const float dc, const int idc, const unsigned sc) {
    const bool better = dc < r.d[K - 1] || (dc == r.d[K - 1] && idc < r.id[K - 1]);
    if (!better) return;
    r.d[K - 1] = dc; r.id[K - 1] = idc; r.slot[K - 1] = sc;
    if (r.found < K) r.found++;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const bool sw = r.d[j] < r.d[j - 1] || (r.d[j] == r.d[j - 1] && r.id[j] < r.id[j - 1]);
        if (sw) {
            const float td = r.d[j]; r.d[j] = r.d[j - 1]; r.d[j - 1] = td;
            const int ti = r.id[j]; r.id[j] = r.id[j - 1]; r.id[j - 1] = ti;
            const unsigned ts = r.slot[j]; r.slot[j] = r.slot[j - 1]; r.slot[j - 1] = ts;
        }
    }
}

template <int K>
__device__ __forceinline__ void knn_scan_cell(const CellGridDev& cgd, const int cx, const int cy, const int cz, const float qx,
                                              const float qy, const float qz, KnnResult<K>& r, unsigned long long& hits,
                                              unsigned long long& cand) {
    const DevGrid& g = cgd.g;
    if (!(abs(cx) < kKeyLimit && abs(cy) < kKeyLimit && abs(cz) < kKeyLimit)) return;
    HashEntry e{kEmptyKey, 0u, 0u};
    if (cgd.win.cells) {  // dense cell window (always present for grids built on the device, which have no hash table)
        const int wx = cx - cgd.win.ox, wy = cy - cgd.win.oy, wz = cz - cgd.win.oz;
        if (!((unsigned)wx < (unsigned)cgd.win.nx && (unsigned)wy < (unsigned)cgd.win.ny && (unsigned)wz < (unsigned)cgd.win.nz)) return;
        const uint2 c = cgd.win.cells[(unsigned)((wz * cgd.win.ny + wy) * cgd.win.nx + wx)];
        if (c.y == 0u) return;
        e.begin = c.x; e.count = c.y;
    } else {
        const unsigned long long key = pack_key(cx, cy, cz);
        unsigned h = hash_key(key) & g.mask;
        e = g.table[h];
        while (e.key != key && e.key != kEmptyKey) {
            h = (h + 1) & g.mask;
            e = g.table[h];
        }
        if (e.key != key) return;
    }
    hits++;
    cand += e.count;
    const unsigned end = e.begin + e.count;
    for (unsigned s = e.begin; s < end; ++s) {
        const float4 p = g.pts[s];
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // flann::L2_Simple<float>
        knn_insert<K>(r, d2, __float_as_int(p.w), s);
    }
}

// gate: squared-distance beyond which the caller does not care (INFINITY = un-gated)
template <int K>
__device__ __forceinline__ void knn_grid(const CellGridDev& cg, const float qx, const float qy, const float qz, const float gate,
                                         KnnResult<K>& r, unsigned long long& probes, unsigned long long& hits,
                                         unsigned long long& cand) {
#pragma unroll
    for (int j = 0; j < K; ++j) { r.d[j] = INFINITY; r.id[j] = 0x7fffffff; r.slot[j] = 0; }
    r.found = 0;
    const double fx = floor((double)qx * cg.inv_cell), fy = floor((double)qy * cg.inv_cell), fz = floor((double)qz * cg.inv_cell);
    if (!(fabs(fx) < (double)kKeyLimit && fabs(fy) < (double)kKeyLimit && fabs(fz) < (double)kKeyLimit)) return;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    bool complete = false;
    for (int rho = 0; rho <= kMaxRing && !complete; ++rho) {
        for (int dz = -rho; dz <= rho; ++dz)
            for (int dy = -rho; dy <= rho; ++dy)
                for (int dx = -rho; dx <= rho; ++dx) {
                    const int m = max(abs(dx), max(abs(dy), abs(dz)));
                    if (m != rho) continue;
                    probes++;
                    knn_scan_cell<K>(cg, cx + dx, cy + dy, cz + dz, qx, qy, qz, r, hits, cand);
                }
        const double rad = (double)rho * cg.cell;
        const double rad2 = rad * rad * (1.0 - 1e-5);
        if (r.found == K && (double)r.d[K - 1] <= rad2) complete = true;
        if (rad2 >= (double)gate * (1.0 + 1e-5)) complete = true;
    }
    if (!complete && cg.win.cells) {
        // Dense cell window: keep walking rings instead of brute-forcing the map (a loop-closure source point whose nearest
        // target lies tens of metres away -- sub-maps that only partly overlap -- used to scan the WHOLE target cloud in one lane).
        // Only the part of a shell that intersects the window is visited, face by face, so all rings together cost at most one visit
        // per window cell; the search ends with the usual bound (K-th distance inside the covered radius) or when the shell has left
        // the window on every side (every cell seen).
        const int wx = cx - cg.win.ox, wy = cy - cg.win.oy, wz = cz - cg.win.oz;  // query cell relative to the window (may be outside)
        const int reach = max(max(max(wx, cg.win.nx - 1 - wx), max(wy, cg.win.ny - 1 - wy)), max(wz, cg.win.nz - 1 - wz));
        for (int rho = kMaxRing + 1; rho <= reach && !complete; ++rho) {
            const int z0 = max(-rho, -wz), z1 = min(rho, cg.win.nz - 1 - wz);
            const int y0 = max(-rho, -wy), y1 = min(rho, cg.win.ny - 1 - wy);
            const int x0 = max(-rho, -wx), x1 = min(rho, cg.win.nx - 1 - wx);
            for (int dz = z0; dz <= z1; ++dz) {
                const bool zface = dz == -rho || dz == rho;
                for (int dy = y0; dy <= y1; ++dy) {
                    const bool face = zface || dy == -rho || dy == rho;
                    if (face) {
                        for (int dx = x0; dx <= x1; ++dx) { probes++; knn_scan_cell<K>(cg, cx + dx, cy + dy, cz + dz, qx, qy, qz, r, hits, cand); }
                    } else {
                        if (-rho >= x0) { probes++; knn_scan_cell<K>(cg, cx - rho, cy + dy, cz + dz, qx, qy, qz, r, hits, cand); }
                        if (rho <= x1) { probes++; knn_scan_cell<K>(cg, cx + rho, cy + dy, cz + dz, qx, qy, qz, r, hits, cand); }
                    }
                }
            }
            const double rad = (double)rho * cg.cell;
            const double rad2 = rad * rad * (1.0 - 1e-5);
            if (r.found == K && (double)r.d[K - 1] <= rad2) complete = true;
            if (rad2 >= (double)gate * (1.0 + 1e-5)) complete = true;
        }
        complete = true;  // rho > reach: every cell of the window has been visited
    }
    if (!complete) {
        // brute force over the whole map (exact by construction; hash-table grids only)
#pragma unroll
        for (int j = 0; j < K; ++j) { r.d[j] = INFINITY; r.id[j] = 0x7fffffff; r.slot[j] = 0; }
        r.found = 0;
        for (unsigned s = 0; s < cg.g.n_pts; ++s) {
            const float4 p = cg.g.pts[s];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            knn_insert<K>(r, (dx * dx + dy * dy) + dz * dz, __float_as_int(p.w), s);
        }
        cand += cg.g.n_pts;
    }
}

// TransformPointCloud(.., Mat4d): R,t cast to float, then r0*x + (r1*y + r2*z) + t in float
// (pointcloud_utility.h:141-158; Eigen un-vectorised 3-term redux order)
struct RtFloat { float R[9]; float t[3]; };
__device__ __forceinline__ RtFloat load_rt_float(const double* T) {
    RtFloat o;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) o.R[i + j * 3] = (float)T[i + j * 4];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = (float)T[12 + i];
    return o;
}
__device__ __forceinline__ void xform_f(const RtFloat& rt, const float x, const float y, const float z, float& ox, float& oy, float& oz) {
    ox = (rt.R[0] * x + (rt.R[3] * y + rt.R[6] * z)) + rt.t[0];
    oy = (rt.R[1] * x + (rt.R[4] * y + rt.R[7] * z)) + rt.t[1];
    oz = (rt.R[2] * x + (rt.R[5] * y + rt.R[8] * z)) + rt.t[2];
}

__device__ __forceinline__ void count_traffic(TrafficCounters* tc, unsigned long long p, unsigned long long h, unsigned long long c) {
    const double sp = wave_sum_u((double)p), sh = wave_sum_u((double)h), sc = wave_sum_u((double)c);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&tc->probes, (unsigned long long)sp);
        atomicAdd(&tc->hits, (unsigned long long)sh);
        atomicAdd(&tc->cand, (unsigned long long)sc);
    }
}

// ---------------------------------------------------------------------------------------------
// 1-NN squared distance of the float-transformed source (fitness loops); d2_out[i] = inf if none
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
nn_dist_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
               const double* __restrict__ T /* 16, device */, const CellGridDev cg, const float gate, float* __restrict__ d2_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const RtFloat rt = load_rt_float(T);
    float qx, qy, qz;
    xform_f(rt, sx[i], sy[i], sz[i], qx, qy, qz);
    KnnResult<1> r;
    unsigned long long a = 0, b = 0, c = 0;
    knn_grid<1>(cg, qx, qy, qz, gate, r, a, b, c);
    d2_out[i] = r.found ? r.d[0] : INFINITY;
}

// ---------------------------------------------------------------------------------------------
// point-to-line on the exact 5-NN of the corner map (Appendix C.2)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool line_residual_dev(const float4 (&nn)[5], const float spx, const float spy, const float spz,
                                                  const float ptx, const float pty, const float ptz, const double* __restrict__ T,
                                                  const double ratio, double (&J)[6], double& dist) {
    double P[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) { P[j][0] = (double)nn[j].x; P[j][1] = (double)nn[j].y; P[j][2] = (double)nn[j].z; }
    double c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = ((((P[0][a] + P[1][a]) + P[2][a]) + P[3][a]) + P[4][a]) / 5.0;
    double D[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int a = 0; a < 3; ++a) D[j][a] = P[j][a] - c[a];
    double C[3][3];  // [col][row]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 5; ++k) s += D[k][i] * D[k][j];
            C[j][i] = s / 5.0;
        }
    double S[3], V[3][3];
    jacobi_svd3_v(C, S, V);
    if (S[0] <= ratio * S[1]) return false;
    const double n0 = V[0][0], n1 = V[0][1], n2 = V[0][2];
    const double ps0 = (double)spx, ps1 = (double)spy, ps2 = (double)spz;
    const double a0 = (double)ptx - c[0], a1 = (double)pty - c[1], a2 = (double)ptz - c[2];
    const double w0 = a1 * n2 - a2 * n1, w1 = a2 * n0 - a0 * n2, w2 = a0 * n1 - a1 * n0;
    dist = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    const double u0 = w0 / dist, u1 = w1 / dist, u2 = w2 / dist;
    const double v0 = (T[0] * ps0 + T[4] * ps1) + T[8] * ps2;
    const double v1 = (T[1] * ps0 + T[5] * ps1) + T[9] * ps2;
    const double v2 = (T[2] * ps0 + T[6] * ps1) + T[10] * ps2;
    // M = SO3Hat(n) * SO3Hat(v)   (column-major 3x3, zero terms kept)
    const double hn[9] = {0.0, n2, -n1, -n2, 0.0, n0, n1, -n0, 0.0};
    const double hv[9] = {0.0, v2, -v1, -v2, 0.0, v0, v1, -v0, 0.0};
    double M[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) M[i + j * 3] = (hn[i] * hv[0 + j * 3] + hn[i + 3] * hv[1 + j * 3]) + hn[i + 6] * hv[2 + j * 3];
#pragma unroll
    for (int i = 0; i < 3; ++i) J[i] = (M[0 + i * 3] * u0 + M[1 + i * 3] * u1) + M[2 + i * 3] * u2;
    // SO3Hat(-n)^T u
    const double hm[9] = {0.0, -n2, n1, n2, 0.0, -n0, -n1, n0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i) J[3 + i] = (hm[0 + i * 3] * u0 + hm[1 + i * 3] * u1) + hm[2 + i * 3] * u2;
    return true;
}

// ---------------------------------------------------------------------------------------------
// IncrementalNDT per-point stage.  Voxel table: key -> voxel slot v (entry.begin), with
// mu[v][3], info[v][9] (column-major), vid[v] (creation id); only estimated voxels are in the table.
// ---------------------------------------------------------------------------------------------
struct NdtGridDev {
    const HashEntry* table;
    unsigned mask;
    const double* mu;    // [nv][3]
    const double* info;  // [nv][9]
    const int* vid;      // [nv]
    double inv_voxel;
};

// incremental image maintenance (matcher_ndt.hpp::sync_image): final table entries and voxel rows of one map update
struct NdtTableEdit { unsigned index, pad; HashEntry e; };
struct NdtRowEdit { unsigned row; int vid; double mu[3]; double info[9]; };
static_assert(sizeof(NdtTableEdit) == 24 && sizeof(NdtRowEdit) == 104, "edit records are packed back to back");

__global__ void __launch_bounds__(256)
ndt_apply_edits_kernel(const NdtTableEdit* __restrict__ te, const int nt, const NdtRowEdit* __restrict__ re, const int nr,
                       HashEntry* __restrict__ table, double* __restrict__ mu, double* __restrict__ info, int* __restrict__ vid) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nt) table[te[t].index] = te[t].e;
    if (t < nr * 4) {
        const NdtRowEdit& r = re[t >> 2];
        const int part = t & 3;
        if (part == 0) {
            vid[r.row] = r.vid;
            mu[3 * (size_t)r.row] = r.mu[0]; mu[3 * (size_t)r.row + 1] = r.mu[1]; mu[3 * (size_t)r.row + 2] = r.mu[2];
        } else {
            const int o = 3 * (part - 1);
            info[9 * (size_t)r.row + o] = r.info[o]; info[9 * (size_t)r.row + o + 1] = r.info[o + 1]; info[9 * (size_t)r.row + o + 2] = r.info[o + 2];
        }
    }
}

template <bool COUNT>
__global__ void __launch_bounds__(64)
ndt_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
           const GnState* __restrict__ st, const int first, const Pose16 T0, const NdtGridDev ng, const double outlier_thr,
           int* __restrict__ hit_vid /* [n][7] */, unsigned char* __restrict__ eff7 /* [n][7] */, double* __restrict__ partials,
           TrafficCounters* __restrict__ tc) {
    const int done = first ? 0 : st->done;
    double P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = first ? T0.m[k] : st->T[k];
    if (done) return;
    const int i = blockIdx.x * 64 + threadIdx.x;
    double Hc[21], Bc[6], res = 0.0, cnt = 0.0;
#pragma unroll
    for (int k = 0; k < 21; ++k) Hc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) Bc[k] = 0.0;
    unsigned long long c_p = 0, c_h = 0, c_c = 0;
    if (i < n) {
        const double p0 = sx[i], p1 = sy[i], p2 = sz[i];
        const double q0 = ((P[0] * p0 + P[4] * p1) + P[8] * p2) + P[12];
        const double q1 = ((P[1] * p0 + P[5] * p1) + P[9] * p2) + P[13];
        const double q2 = ((P[2] * p0 + P[6] * p1) + P[10] * p2) + P[14];
        // (point_transformed * inv_voxel_size_).cast<int>(): truncation toward zero (incremental_ndt.h:256)
        const double f0 = q0 * ng.inv_voxel, f1 = q1 * ng.inv_voxel, f2 = q2 * ng.inv_voxel;
        const bool in_range = fabs(f0) < (double)kKeyLimit && fabs(f1) < (double)kKeyLimit && fabs(f2) < (double)kKeyLimit;
        const int kx = in_range ? (int)f0 : 0, ky = in_range ? (int)f1 : 0, kz = in_range ? (int)f2 : 0;
        // J = [ -R*SO3Hat(p) | I ]
        const double hat[9] = {0.0, p2, -p1, -p2, 0.0, p0, p1, -p0, 0.0};
        double J[18];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                J[q + j * 3] = -((P[q] * hat[0 + j * 3] + P[q + 4] * hat[1 + j * 3]) + P[q + 8] * hat[2 + j * 3]);
                J[q + (j + 3) * 3] = (q == j) ? 1.0 : 0.0;
            }
        const int nb[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, 1}};  // :122-127
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            int vid = -1;
            unsigned char ef = 0;
            if (in_range) {
                const unsigned long long key = pack_key(kx + nb[k][0], ky + nb[k][1], kz + nb[k][2]);
                unsigned h = hash_key(key) & ng.mask;
                HashEntry e = ng.table[h];
                while (e.key != key && e.key != kEmptyKey) { h = (h + 1) & ng.mask; e = ng.table[h]; }
                if (COUNT) c_p++;
                if (e.key == key && e.count != 0u) {  // count == 0: a voxel that exists but has no estimate yet (device map update)
                    if (COUNT) c_h++;
                    const unsigned v = e.begin;
                    const double e0 = q0 - ng.mu[3 * v], e1 = q1 - ng.mu[3 * v + 1], e2 = q2 - ng.mu[3 * v + 2];
                    double I[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) I[q] = ng.info[9 * (size_t)v + q];
                    const double ei0 = (e0 * I[0] + e1 * I[1]) + e2 * I[2];
                    const double ei1 = (e0 * I[3] + e1 * I[4]) + e2 * I[5];
                    const double ei2 = (e0 * I[6] + e1 * I[7]) + e2 * I[8];
                    const double r = (ei0 * e0 + ei1 * e1) + ei2 * e2;
                    if (!(r != r) && !(r > outlier_thr)) {
                        double JtI[18];  // 6x3 col-major
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int a = 0; a < 6; ++a)
                                JtI[a + c * 6] = (J[0 + a * 3] * I[0 + c * 3] + J[1 + a * 3] * I[1 + c * 3]) + J[2 + a * 3] * I[2 + c * 3];
                        int kk = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int b = a; b < 6; ++b) {
                                Hc[kk] += (JtI[a + 0 * 6] * J[0 + b * 3] + JtI[a + 1 * 6] * J[1 + b * 3]) + JtI[a + 2 * 6] * J[2 + b * 3];
                                ++kk;
                            }
#pragma unroll
                        for (int a = 0; a < 6; ++a) Bc[a] += ((-JtI[a + 0 * 6]) * e0 + (-JtI[a + 1 * 6]) * e1) + (-JtI[a + 2 * 6]) * e2;
                        res += r;
                        cnt += 1.0;
                        ef = 1;
                        vid = ng.vid[v];
                    }
                }
            }
            hit_vid[(size_t)i * 7 + k] = vid;
            eff7[(size_t)i * 7 + k] = ef;
        }
    }
    const int lane = threadIdx.x & 63;
    double* row = partials + (size_t)blockIdx.x * kPartialStride;
#pragma unroll
    for (int k = 0; k < 21; ++k) { const double v = wave_sum_dpp(Hc[k]); if (lane == 63) row[k] = v; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double v = wave_sum_dpp(Bc[k]); if (lane == 63) row[21 + k] = v; }
    const double sr = wave_sum_dpp(res), sc = wave_sum_dpp(cnt);
    if (lane == 63) { row[27] = sr; row[28] = sc; }
    if (COUNT) count_traffic(tc, c_p, c_h, c_c);
}

// ---------------------------------------------------------------------------------------------
// ICP / NDT Gauss-Newton tail.  mode 0 = IcpOptimized (state [dt, dtheta], det==0 skip, converged
// flag), mode 1 = IncrementalNDT (state [dtheta, dt], min_effective early-out).  Both right-multiply.
// ---------------------------------------------------------------------------------------------
struct LuTailSmem {
    double red[32][33];
    double tot[32];
    double Hs[36], inv[36], gs[6], xs[6];
    int tr[6];
};
struct LuTailArgs {  // what the tail needs besides the rows (one struct so that the fused kernels stay readable)
    int mode;            // 0 = IcpOptimized, 1 = IncrementalNDT
    double rot_thr, pos_thr;
    int min_effective;
    Mailbox* mb;
    unsigned match_id;   // launch word: max_iterations << 24 | exact-solver flag << 23 | match id
};
// executed by a whole workgroup of NT threads (all reduce the rows, wave 0 solves); SC1: the rows were published write-through by
// other workgroups of the SAME launch (fused search + fit + tail kernels)
template <int NT, bool SC1>
__device__ __forceinline__ void lu_tail(GnState* __restrict__ st, LuTailSmem& sm, const double* __restrict__ partials, const int nrows, const LuTailArgs a,
                                        double (&Tl)[16], const int it) {
    double (&red)[32][33] = sm.red;
    double (&tot)[32] = sm.tot;
    double (&Hs)[36] = sm.Hs;
    double (&inv)[36] = sm.inv;
    double (&gs)[6] = sm.gs;
    double (&xs)[6] = sm.xs;
    int (&tr)[6] = sm.tr;
    const int mode = a.mode, min_effective = a.min_effective;
    const double rot_thr = a.rot_thr, pos_thr = a.pos_thr;
    Mailbox* const mb = a.mb;
    const unsigned match_id = a.match_id;
    reduce_partials<NT, SC1, (NT <= 512 ? 32 : 16)>(partials, nrows, tot, red);  // (one trip for the fused kernels' 205 / 457 rows)
    FLS_STAMP(3);
    if (threadIdx.x >= 64) return;  // wave 0 only
    const int lane = threadIdx.x;
    if (lane < 36) {
        const int i = lane % 6, j = lane / 6;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const int k = a * 6 - (a * (a - 1)) / 2 + (b - a);
        Hs[lane] = tot[k];
        st->H[lane] = tot[k];
    }
    if (lane < 6) { gs[lane] = tot[21 + lane]; st->g[lane] = tot[21 + lane]; }
    __builtin_amdgcn_wave_barrier();
    const int effective = (int)tot[28];
    const double sres = tot[27];
    const bool early_fail = (mode == 1 && effective < min_effective);  // incremental_ndt.h:306-309: T = pose; return false
    double det = 1.0;
    if (!early_fail) {
        // SPD fast path (kernels_p2plane.hpp::ldlt_solve6_lane): positive pivots imply det(H) > 0, so the reference's exact
        // det == 0 test (icp_optimized.h:129, Q14) cannot fire; anything else goes through the restated LU inverse
        const int fast = ldlt_fast_path(Hs, gs, xs, match_id);
        if (!fast) det = lu6_solve_wave(Hs, inv, gs, xs, tr);
    }
    FLS_STAMP(4);
    if (lane == 0) {
        st->n_valid = effective;
        st->sum_res = sres;
        int stop = 0, conv = 0;
        double dxo[6] = {0, 0, 0, 0, 0, 0};
        if (early_fail) {
            stop = 1;
        } else if (mode == 0 && det == 0.0) {
            // icp_optimized.h:129-131: skip the update, keep iterating
        } else {
            double dx[6];
            for (int q = 0; q < 6; ++q) dx[q] = xs[q];
            const double* dth = (mode == 0) ? dx + 3 : dx;
            const double* dt = (mode == 0) ? dx : dx + 3;
            double Rd[9], R[9], Rn[9];
            if (mode == 0) { Tl[12] += dt[0]; Tl[13] += dt[1]; Tl[14] += dt[2]; }
            so3_exp_dev(dth, Rd);
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + j * 3] = Tl[i + j * 4];
            mat3_mul_dev(R, Rd, Rn);  // right-multiplicative
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) Tl[i + j * 4] = Rn[i + j * 3];
            if (mode == 1) { Tl[12] += dt[0]; Tl[13] += dt[1]; Tl[14] += dt[2]; }
            for (int q = 0; q < 6; ++q) { st->last_dx[q] = dx[q]; dxo[q] = dx[q]; }
            if (norm3d(dth) < rot_thr && norm3d(dt) < pos_thr) { stop = 1; conv = 1; }
        }
        for (int q = 0; q < 16; ++q) st->T[q] = Tl[q];
        st->done = stop;
        st->converged = conv;
        if (it < kMaxIter) {
            for (int q = 0; q < 16; ++q) st->log_T[it][q] = Tl[q];
            st->log_nv[it] = effective;
            st->log_res[it] = sres;
        }
        st->iter = it + 1;
        if (mb) {
            mailbox_publish(mb, Tl, dxo, sres, 0.0, it + 1, stop, conv, effective, 0, match_id);  // launch word: max_iterations << 24 | match id
        }
        FLS_STAMP(5);
    }
}

// ndt_lanes_kernel: the same per-point lambda with ONE LANE PER NEIGHBOUR VOXEL (8 lanes per point, lane 7 idle): the seven
// dependent look-up chains of a point (table probe -> mean / information -> Mahalanobis gate -> J^T Sigma^-1 J) run side by side
// and the launch has eight times the waves of ndt_kernel (which runs 29k points as 457 single-wave workgroups, under one wave
// per SIMD).  512 threads = 64 points per workgroup: the number of partial rows stays what gn_solve_lu_kernel reads in one pass.
// Per point the seven contributions are now added by the wave reduction tree instead of sequentially (last-bit differences in H, g;
// counts and flags are integers).
constexpr int kNdtLanesBlock = 512;
// FUSED = the Gauss-Newton tail in the last workgroup (default since round 6, FLS_FUSED_TAIL=0 for the separate gn_solve_lu_kernel launch; see
// matcher_ndt.hpp::fused_tail for why it was slower until the pose moved to LDS).  A template parameter, not a run-time branch.
template <bool FUSED>
__global__ void __launch_bounds__(kNdtLanesBlock)
ndt_lanes_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                 GnState* __restrict__ st, const int first, const Pose16 T0, const NdtGridDev ng, const double outlier_thr,
                 int* __restrict__ hit_vid /* [n][7] */, unsigned char* __restrict__ eff7 /* [n][7] */, double* __restrict__ partials,
                 unsigned* __restrict__ ticket /* nullptr: the tail runs as its own launch */, const int shards, const LuTailArgs tail) {
    const int done = first ? 0 : st->done;
    double P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = first ? T0.m[k] : st->T[k];
    const int it = (FUSED && !first) ? st->iter : 0;
    if (done) return;
    __shared__ double wsum[kNdtLanesBlock / 64][32];
    __shared__ double s_pose[16];
    if (FUSED && threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) s_pose[q] = P[q];
    }
    const int i = (blockIdx.x * kNdtLanesBlock + threadIdx.x) >> 3, k = threadIdx.x & 7;
    double Hc[21], Bc[6], res = 0.0, cnt = 0.0;
#pragma unroll
    for (int q = 0; q < 21; ++q) Hc[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) Bc[q] = 0.0;
    if (i < n && k < 7) {
        const double p0 = sx[i], p1 = sy[i], p2 = sz[i];
        const double q0 = ((P[0] * p0 + P[4] * p1) + P[8] * p2) + P[12];
        const double q1 = ((P[1] * p0 + P[5] * p1) + P[9] * p2) + P[13];
        const double q2 = ((P[2] * p0 + P[6] * p1) + P[10] * p2) + P[14];
        const double f0 = q0 * ng.inv_voxel, f1 = q1 * ng.inv_voxel, f2 = q2 * ng.inv_voxel;
        const bool in_range = fabs(f0) < (double)kKeyLimit && fabs(f1) < (double)kKeyLimit && fabs(f2) < (double)kKeyLimit;
        const int kx = in_range ? (int)f0 : 0, ky = in_range ? (int)f1 : 0, kz = in_range ? (int)f2 : 0;
        // neighbour k of {0, -x, +x, +y, -y, -z, +z} (:122-127), branch-free
        const int ox = k == 1 ? -1 : (k == 2 ? 1 : 0), oy = k == 3 ? 1 : (k == 4 ? -1 : 0), oz = k == 5 ? -1 : (k == 6 ? 1 : 0);
        int vid = -1;
        unsigned char ef = 0;
        if (in_range) {
            const unsigned long long key = pack_key(kx + ox, ky + oy, kz + oz);
            unsigned h = hash_key(key) & ng.mask;
            HashEntry e = ng.table[h];
            while (e.key != key && e.key != kEmptyKey) { h = (h + 1) & ng.mask; e = ng.table[h]; }
            if (e.key == key && e.count != 0u) {
                const unsigned v = e.begin;
                const double e0 = q0 - ng.mu[3 * v], e1 = q1 - ng.mu[3 * v + 1], e2 = q2 - ng.mu[3 * v + 2];
                double I[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) I[q] = ng.info[9 * (size_t)v + q];
                const double ei0 = (e0 * I[0] + e1 * I[1]) + e2 * I[2];
                const double ei1 = (e0 * I[3] + e1 * I[4]) + e2 * I[5];
                const double ei2 = (e0 * I[6] + e1 * I[7]) + e2 * I[8];
                const double r = (ei0 * e0 + ei1 * e1) + ei2 * e2;
                if (!(r != r) && !(r > outlier_thr)) {
                    const double hat[9] = {0.0, p2, -p1, -p2, 0.0, p0, p1, -p0, 0.0};
                    double J[18];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            J[q + j * 3] = -((P[q] * hat[0 + j * 3] + P[q + 4] * hat[1 + j * 3]) + P[q + 8] * hat[2 + j * 3]);
                            J[q + (j + 3) * 3] = (q == j) ? 1.0 : 0.0;
                        }
                    double JtI[18];  // 6x3 col-major
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int a = 0; a < 6; ++a)
                            JtI[a + c * 6] = (J[0 + a * 3] * I[0 + c * 3] + J[1 + a * 3] * I[1 + c * 3]) + J[2 + a * 3] * I[2 + c * 3];
                    int kk = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = a; b < 6; ++b) {
                            Hc[kk] = (JtI[a + 0 * 6] * J[0 + b * 3] + JtI[a + 1 * 6] * J[1 + b * 3]) + JtI[a + 2 * 6] * J[2 + b * 3];
                            ++kk;
                        }
#pragma unroll
                    for (int a = 0; a < 6; ++a) Bc[a] = ((-JtI[a + 0 * 6]) * e0 + (-JtI[a + 1 * 6]) * e1) + (-JtI[a + 2 * 6]) * e2;
                    res = r;
                    cnt = 1.0;
                    ef = 1;
                    vid = ng.vid[v];
                }
            }
        }
        hit_vid[(size_t)i * 7 + k] = vid;
        eff7[(size_t)i * 7 + k] = ef;
    }
    const int lane = threadIdx.x & 63;
    double* row = &wsum[threadIdx.x >> 6][0];
#pragma unroll
    for (int q = 0; q < 21; ++q) { const double v = wave_sum_dpp(Hc[q]); if (lane == 63) row[q] = v; }
#pragma unroll
    for (int q = 0; q < 6; ++q) { const double v = wave_sum_dpp(Bc[q]); if (lane == 63) row[21 + q] = v; }
    const double sr = wave_sum_dpp(res), sc = wave_sum_dpp(cnt);
    if (lane == 63) { row[27] = sr; row[28] = sc; }
    __syncthreads();
    double v = 0.0;
    if (threadIdx.x < 29) {
#pragma unroll
        for (int w = 0; w < kNdtLanesBlock / 64; ++w) v += wsum[w][threadIdx.x];
        if (!FUSED) partials[(size_t)blockIdx.x * kPartialStride + threadIdx.x] = v;
    }
    if constexpr (FUSED) {
        // fused Gauss-Newton tail (round 3): the row goes out write-through, the last workgroup to arrive solves and publishes
        __shared__ unsigned s_ticket;
        __shared__ LuTailSmem sm;
        if (!publish_row_and_arrive(v, threadIdx.x < 29, partials, ticket, shards, s_ticket)) return;
        double Tl[16];  // (the pose parked in LDS at the start: sixteen doubles kept in registers across the per-point part cost the kernel its fourth wave per SIMD)
#pragma unroll
        for (int q = 0; q < 16; ++q) Tl[q] = s_pose[q];
        lu_tail<kNdtLanesBlock, true>(st, sm, partials, (int)gridDim.x, tail, Tl, it);
    }
}

__global__ void __launch_bounds__(1024)
gn_solve_lu_kernel(GnState* __restrict__ st, const int first, const Pose16 T0, const double* __restrict__ partials, const int nrows,
                   const int mode, const double rot_thr, const double pos_thr, const int min_effective, Mailbox* __restrict__ mb,
                   const unsigned match_id) {
    const int done = first ? 0 : st->done;
    double Tl[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Tl[q] = first ? T0.m[q] : st->T[q];
    const int it = first ? 0 : st->iter;
    if (done) return;
    __shared__ LuTailSmem sm;
    lu_tail<kSolveThreads, false>(st, sm, partials, nrows, LuTailArgs{mode, rot_thr, pos_thr, min_effective, mb, match_id}, Tl, it);
}

}  // namespace fls
