// loop_closure.hpp -- host side of fls_loop_match, the replacement of LoopClosure::Match (src/slam/loop_closure.cpp:233-267):
//
//     static const std::vector<float> resolution{10.0, 5.0, 3.0, 2.0};
//     pcl::NormalDistributionsTransform ndt;  setStepSize(0.5)  setMaximumIterations(30)           :239-242
//     for r: setResolution(r); source / target = VoxelGridCloud(.., r * 0.2f); align(pose)          :244-252
//     pcl::GeneralizedIterativeClosestPoint gicp over VoxelGridCloud(source, 0.5f) / (target, 0.4f),
//         setMaximumIterations(30) setMaxCorrespondenceDistance(2.0); align(pose)                    :254-263
//     return gicp.getFitnessScore()                                                                  :265
//
// PCL is a third-party dependency that is not under /root/reference; what is implemented is the published algorithm each PCL 1.10
// class implements, with the parameters the reference sets and PCL's defaults for the rest:
//   NormalDistributionsTransform (ndt.hpp)       Magnusson 2009 P2D-NDT: leaf Gaussians of VoxelGridCovariance (>= 6 points, eigenvalue
//                                                floor 0.01), outlier ratio 0.55, Newton step from a 6x6 SVD solve, More-Thuente line
//                                                search (mu 1e-4, nu 0.9, <= 10 trials, step in [epsilon / 2, step size]), stop when the
//                                                step length drops below transformation epsilon 0.1
//   GeneralizedIterativeClosestPoint (gicp.hpp)  Segal 2009: 20-NN covariances with singular values (1, 1, 1e-3), nearest-neighbour
//                                                correspondences inside the gate, BFGS (GSL vector_bfgs2, <= 20 inner iterations,
//                                                gradient tolerance 1e-2) on sum res^T (C2 + R C1 R^T)^-1 res, stop on rotation 2e-3 /
//                                                translation 5e-4 change
// Division of labour: everything per POINT runs on the device (kernels_loop.hpp: score / gradient / Hessian sums, 20-NN covariances,
// correspondences + Mahalanobis matrices, cost + gradient sums, fitness); the six-parameter optimisers and the leaf statistics of
// the (already down-sampled) target run on the host, which learns every reduction result through a host-mapped block (no stream
// synchronisation per evaluation).  The VoxelGridCloud calls use the exact host filter by default, the device filter with
// FLS_DEVICE_VOXELGRID=1.
#pragma once
#include <chrono>
#include "device_voxelgrid.hpp"
#include "kernels_loop.hpp"
#include "fitness_host.hpp"
#include <map>
#include <thread>
#include <mutex>
#include <exception>
#include <condition_variable>

namespace fls {

struct LoopMatcher {
    std::mutex run_mx;  // held by fls_loop_match for the duration of a Match
    int device = 0;
    hipStream_t stream = nullptr;
    LoopMail* mail_host = nullptr;
    LoopMail* mail_dev = nullptr;
    unsigned seq = 0;
    DevBuf<double> d_rows;
    DevBuf<unsigned> d_ticket;
    bool fused_reduce = true;  // FLS_LOOP_FUSED_REDUCE=0: partial rows summed by a launch of their own
    bool device_filter = false;
    bool debug = false;  // FLS_LOOP_DEBUG=1: one line per GICP outer iteration on stderr
    fls_loop_stats st{};

    ~LoopMatcher() {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (mail_host) (void)hipHostFree(mail_host);
    }
    void init(int dev) {
        device = dev;
        FLS_HIP(hipSetDevice(device));
        FLS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        FLS_HIP(hipHostMalloc((void**)&mail_host, sizeof(LoopMail), hipHostMallocMapped));
        std::memset(mail_host, 0, sizeof(LoopMail));
        FLS_HIP(hipHostGetDevicePointer((void**)&mail_dev, mail_host, 0));
        read_env();
    }
    void read_env() {  // per Match: one matcher per device is kept alive across calls (fls_reg.hip)
        device_filter = false; debug = false; fused_reduce = true;
        if (const char* e = std::getenv("FLS_LOOP_FUSED_REDUCE")) fused_reduce = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_DEVICE_VOXELGRID")) device_filter = std::atoi(e) != 0;
        if (const char* e = std::getenv("FLS_LOOP_DEBUG")) debug = std::atoi(e) != 0;
    }

    // ---- float pose algebra (Eigen::Transform<float, 3, Affine>) ------------------------------------------------------------
    static LoopMat4f ident() { LoopMat4f r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
    static LoopMat4f mul(const LoopMat4f& a, const LoopMat4f& b) {
        LoopMat4f r;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i)
                r.m[i + 4 * j] = ((a.m[i] * b.m[4 * j] + a.m[i + 4] * b.m[1 + 4 * j]) + a.m[i + 8] * b.m[2 + 4 * j]) + a.m[i + 12] * b.m[3 + 4 * j];
        return r;
    }
    static LoopMat4f rot(int axis, float angle) {  // AngleAxisf(angle, unit axis)
        LoopMat4f r = ident();
        const float c = std::cos(angle), s = std::sin(angle);
        const int a = (axis + 1) % 3, b = (axis + 2) % 3;
        r.m[a + 4 * a] = c; r.m[b + 4 * b] = c; r.m[b + 4 * a] = s; r.m[a + 4 * b] = -s;
        return r;
    }
    static LoopMat4f trans(float x, float y, float z) { LoopMat4f r = ident(); r.m[12] = x; r.m[13] = y; r.m[14] = z; return r; }
    static LoopMat4f from_d(const double* T) { LoopMat4f r; for (int i = 0; i < 16; ++i) r.m[i] = float(T[i]); return r; }
    // (Translation(p0..2) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z)).matrix()   ndt.hpp computeStepLengthMT
    static LoopMat4f pose_xyz(const double* p) {
        return mul(mul(mul(trans(float(p[0]), float(p[1]), float(p[2])), rot(0, float(p[3]))), rot(1, float(p[4]))), rot(2, float(p[5])));
    }
    // Matrix3f::eulerAngles(0, 1, 2) (Eigen 3.3 EulerAngles.h)
    static void euler012(const LoopMat4f& t, float (&e)[3]) {
        auto at = [&](int i, int j) { return t.m[i + 4 * j]; };
        const float pi = 3.14159265358979323846f;
        e[0] = std::atan2(at(1, 2), at(2, 2));
        const float c2 = std::sqrt(at(0, 0) * at(0, 0) + at(0, 1) * at(0, 1));
        if (e[0] > 0.f) { e[0] -= pi; e[1] = std::atan2(-at(0, 2), -c2); }
        else e[1] = std::atan2(-at(0, 2), c2);
        const float s1 = std::sin(e[0]), c1 = std::cos(e[0]);
        e[2] = std::atan2(s1 * at(2, 0) - c1 * at(1, 0), c1 * at(1, 1) - s1 * at(2, 1));
        e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2];
    }

    // ---- reductions: launch `fill` (writes nrows partial rows), sum them on the device, wait for the host-mapped block ---------
    template <class F>
    const double* reduce(int nrows, int nv, F&& fill) {
        d_rows.reserve(size_t(std::max(nrows, 1)) * kLoopMaxV);
        if (!d_ticket.p) {
            d_ticket.reserve(kTicketWords);
            FLS_HIP(hipMemsetAsync(d_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));
        }
        seq = (seq + 1u) & 0x7fffffffu;
        if (seq == 0u) seq = 1u;
        fill(LoopOut{d_rows.p, fused_reduce ? d_ticket.p : nullptr, mail_dev, seq});
        if (!fused_reduce) hipLaunchKernelGGL(loop_reduce_kernel, dim3(1), dim3(64), 0, stream, (const double*)d_rows.p, nrows, nv, mail_dev, seq);
        FLS_HIP(hipGetLastError());
        for (unsigned long long spin = 1;; ++spin) {
            if (__atomic_load_n(&mail_host->seq, __ATOMIC_ACQUIRE) == seq) break;
            if ((spin & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) FLS_HIP(q);
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        // (the stream went idle: the block must be there -- anything else is a lost fan-in, never a result to use)
        if (__atomic_load_n(&mail_host->seq, __ATOMIC_ACQUIRE) != seq) throw HipError(hipErrorUnknown, "fls_loop_match: the device left no result block for an evaluation");
        return mail_host->v;
    }

    // a cloud as x | y | z planes on the device
    struct DevCloud {
        DevBuf<float> xyz;
        PinnedBuf<float> stage;
        size_t n = 0;
        const float* x() const { return xyz.p; }
        const float* y() const { return xyz.p + n; }
        const float* z() const { return xyz.p + 2 * n; }
        void upload(const std::vector<PtI>& c, hipStream_t s) {
            n = c.size();
            if (!n) return;
            stage.reserve(3 * n);
            xyz.reserve(3 * n);
            for (size_t i = 0; i < n; ++i) { stage.p[i] = c[i].x; stage.p[n + i] = c[i].y; stage.p[2 * n + i] = c[i].z; }
            FLS_HIP(hipMemcpyAsync(xyz.p, stage.p, 3 * n * sizeof(float), hipMemcpyHostToDevice, s));
            FLS_HIP(hipStreamSynchronize(s));
        }
    };

    DevScan raw;  // (device filter's staging + kernels' scratch: kept across calls)
    DeviceVoxelGrid vg;
    std::vector<PtI> filter(const std::vector<PtI>& c, float leaf) {  // VoxelGridCloud (pointcloud_utility.h:216-271)
        if (device_filter && !c.empty()) {
            raw.upload_raw(&c[0].x, c.size(), 4, stream, true);
            if (vg.run(raw.x.p, raw.y.p, raw.z.p, raw.xyz.p + 3 * c.size(), c.size(), leaf, stream)) {
                std::vector<float> tmp;
                return vg.download(stream, tmp);
            }
        }
        return voxel_grid(c, leaf);
    }

    // ======================================================================================================================
    // NormalDistributionsTransform
    // ======================================================================================================================
    struct TargetLeaves {  // VoxelGridCovariance(leaf = resolution, min_points_per_voxel 6, min_covar_eigvalue_mult 0.01), searchable
        int min_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0};
        float inv = 1.f;
        std::vector<int> leaf_row;
        std::vector<double> mean, icov;
        std::vector<float> centroid;
        size_t rows = 0;
        DevBuf<int> d_leaf_row;
        DevBuf<double> d_mean, d_icov;
        DevBuf<float> d_centroid;
        // sparse form (grid volume above kMaxDenseLeafCells: a stray far point must not cost gigabytes): cell -> row through an
        // open-addressing table of {cell, row} pairs instead of the dense [div2][div1][div0] table
        std::vector<int2> leaf_hash;
        DevBuf<int2> d_leaf_hash;
        unsigned hash_mask = 0;
        bool sparse = false;
    };
    static constexpr size_t kMaxDenseLeafCells = size_t(32) << 20;  // 32 Mi cells: 128 MiB per dense int table (host x 2, device x 1)
    static unsigned leaf_hash_of(const int cell) { return unsigned(cell) * 2654435761u; }
    // release host / device tables an outlier call inflated (the matcher is cached for the life of the process)
    template <class V> static void shrink_if_oversized(V& v, const size_t need) {
        if (v.capacity() > (size_t(4) << 20) && v.capacity() > 4 * need) { V().swap(v); }
    }
    template <class T> static void shrink_if_oversized(DevBuf<T>& b, const size_t need) {
        if (b.cap > (size_t(4) << 20) && b.cap > 4 * need) { (void)hipDeviceSynchronize(); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    }
    struct LeafAcc { int n = 0; double sum[3] = {0, 0, 0}, xx[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; float csum[3] = {0, 0, 0}; };
    std::vector<int> acc_of;
    std::vector<LeafAcc> accs;

    // symmetric 3x3 eigen-decomposition, ascending (SelfAdjointEigenSolver stand-in: Jacobi via hm::svd3)
    static void sym_eig3(const double* A, double* ev, double* evec) {
        double U[9], S[3], V[9];
        hm::svd3(A, U, S, V);
        for (int k = 0; k < 3; ++k) {
            const double sgn = (U[3 * k] * V[3 * k] + U[3 * k + 1] * V[3 * k + 1]) + U[3 * k + 2] * V[3 * k + 2];
            const int dst = 2 - k;
            ev[dst] = sgn < 0.0 ? -S[k] : S[k];
            for (int i = 0; i < 3; ++i) evec[i + 3 * dst] = V[i + 3 * k];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = a + 1; b < 3; ++b)
                if (ev[b] < ev[a]) { std::swap(ev[a], ev[b]); for (int i = 0; i < 3; ++i) std::swap(evec[i + 3 * a], evec[i + 3 * b]); }
    }

    bool build_target(const std::vector<PtI>& in, float resolution, TargetLeaves& tl) {
        tl.rows = 0;
        tl.inv = 1.0f / resolution;
        const float inv = tl.inv;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool any = false;
        for (const PtI& p : in) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            any = true;
            mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
            mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
            mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
        }
        if (!any) return false;
        const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > (long long)INT_MAX) return false;
        for (int a = 0; a < 3; ++a) { tl.min_b[a] = int(std::floor(mn[a] * inv)); tl.div_b[a] = int(std::floor(mx[a] * inv)) - tl.min_b[a] + 1; }
        const int m1 = tl.div_b[0], m2 = tl.div_b[0] * tl.div_b[1];
        // PCL keeps the leaves in a std::map (ascending leaf index) and adds a leaf's points in cloud order; the same sums and the same
        // emission order come from a dense cell -> accumulator table (the grid volume is bounded above) walked in ascending cell index.
        // (A std::map here cost ~100 ns per point: 4-5 ms per Match.)
        const size_t n_cells = size_t(tl.div_b[0]) * tl.div_b[1] * tl.div_b[2];
        tl.sparse = n_cells > kMaxDenseLeafCells;
        accs.clear();
        auto accumulate = [](LeafAcc& l, const PtI& p) {
            const double q[3] = {double(p.x), double(p.y), double(p.z)};
            for (int a = 0; a < 3; ++a) l.sum[a] += q[a];
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) l.xx[r + 3 * c] += q[r] * q[c];
            l.csum[0] += p.x; l.csum[1] += p.y; l.csum[2] += p.z;
            ++l.n;
        };
        auto cell_of = [&](const PtI& p) {
            const int i0 = int(std::floor(p.x * inv) - float(tl.min_b[0])), i1 = int(std::floor(p.y * inv) - float(tl.min_b[1])), i2 = int(std::floor(p.z * inv) - float(tl.min_b[2]));
            return i0 + i1 * m1 + i2 * m2;  // < INT_MAX (checked above)
        };
        // occupied cells in ascending cell index with their accumulators: `order` = {cell, accumulator}
        std::vector<std::pair<int, int>> order;
        if (!tl.sparse) {
            shrink_if_oversized(acc_of, n_cells);
            acc_of.assign(n_cells, -1);
            for (const PtI& p : in) {
                if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
                int& slot = acc_of[size_t(cell_of(p))];
                if (slot < 0) { slot = int(accs.size()); accs.emplace_back(); }
                accumulate(accs[size_t(slot)], p);
            }
            order.reserve(accs.size());
            for (size_t cell_i = 0; cell_i < n_cells; ++cell_i)
                if (acc_of[cell_i] >= 0) order.emplace_back(int(cell_i), acc_of[cell_i]);
        } else {
            // sparse: O(points) memory whatever the extent.  (cell, point) pairs sorted: ascending cell, a cell's points in cloud order --
            // the same sums in the same order as the dense table gives
            std::vector<std::pair<int, unsigned>> cp;
            cp.reserve(in.size());
            for (size_t k = 0; k < in.size(); ++k) {
                const PtI& p = in[k];
                if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
                cp.emplace_back(cell_of(p), unsigned(k));
            }
            std::sort(cp.begin(), cp.end());
            for (size_t a = 0; a < cp.size();) {
                size_t b = a;
                accs.emplace_back();
                for (; b < cp.size() && cp[b].first == cp[a].first; ++b) accumulate(accs.back(), in[cp[b].second]);
                order.emplace_back(cp[a].first, int(accs.size()) - 1);
                a = b;
            }
            std::vector<int>().swap(acc_of);
        }
        if (!tl.sparse) { shrink_if_oversized(tl.leaf_row, n_cells); tl.leaf_row.assign(n_cells, -1); }
        else {
            std::vector<int>().swap(tl.leaf_row);
            size_t hs = 1024;
            while (hs < 2 * order.size() + 2) hs <<= 1;
            tl.hash_mask = unsigned(hs - 1);
            tl.leaf_hash.assign(hs, make_int2(-1, -1));
        }
        tl.mean.clear(); tl.icov.clear(); tl.centroid.clear();
        for (const std::pair<int, int>& oc : order) {
            const size_t cell_i = size_t(oc.first);
            const LeafAcc& l = accs[size_t(oc.second)];
            const int n = l.n;
            if (n < 6) continue;
            double mean[3], cov[9], icov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int a = 0; a < 3; ++a) mean[a] = l.sum[a] / n;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) cov[r + 3 * c] = (l.xx[r + 3 * c] - 2.0 * (l.sum[r] * mean[c])) / n + mean[r] * mean[c];
            for (int i = 0; i < 9; ++i) cov[i] *= (n - 1.0) / n;
            double ev[3], evec[9];
            sym_eig3(cov, ev, evec);
            if (!(ev[0] < 0 || ev[1] < 0 || ev[2] <= 0)) {  // (else: the leaf stays searchable with a zero inverse covariance, as in PCL)
                const double floor_ev = 0.01 * ev[2];
                if (ev[0] < floor_ev) {
                    ev[0] = floor_ev;
                    if (ev[1] < floor_ev) ev[1] = floor_ev;
                    double vinv[9], tmp[9];
                    hm::inv3(evec, vinv);
                    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) tmp[r + 3 * c] = evec[r + 3 * c] * ev[c];
                    hm::mul3(tmp, vinv, cov);
                }
                hm::inv3(cov, icov);
            }
            if (!tl.sparse) tl.leaf_row[cell_i] = int(tl.rows);
            else {
                unsigned h = leaf_hash_of(int(cell_i)) & tl.hash_mask;
                while (tl.leaf_hash[h].x >= 0) h = (h + 1) & tl.hash_mask;
                tl.leaf_hash[h] = make_int2(int(cell_i), int(tl.rows));
            }
            for (int a = 0; a < 3; ++a) { tl.mean.push_back(mean[a]); tl.centroid.push_back(l.csum[a] / float(n)); }
            for (int a = 0; a < 9; ++a) tl.icov.push_back(icov[a]);
            ++tl.rows;
        }
        if (tl.rows == 0) return false;
        shrink_if_oversized(tl.d_leaf_row, tl.leaf_row.size());
        tl.d_mean.reserve(tl.mean.size()); tl.d_icov.reserve(tl.icov.size()); tl.d_centroid.reserve(tl.centroid.size());
        if (!tl.sparse) {
            tl.d_leaf_row.reserve(tl.leaf_row.size());
            FLS_HIP(hipMemcpyAsync(tl.d_leaf_row.p, tl.leaf_row.data(), tl.leaf_row.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        } else {
            tl.d_leaf_hash.reserve(tl.leaf_hash.size());
            FLS_HIP(hipMemcpyAsync(tl.d_leaf_hash.p, tl.leaf_hash.data(), tl.leaf_hash.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
        }
        FLS_HIP(hipMemcpyAsync(tl.d_mean.p, tl.mean.data(), tl.mean.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        FLS_HIP(hipMemcpyAsync(tl.d_icov.p, tl.icov.data(), tl.icov.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        FLS_HIP(hipMemcpyAsync(tl.d_centroid.p, tl.centroid.data(), tl.centroid.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        return true;
    }

    struct NdtRun {
        float resolution = 1.f;
        double step_size = 0.5, outlier_ratio = 0.55, transformation_epsilon = 0.1;
        int max_iterations = 30;
        const DevCloud* src = nullptr;
        const TargetLeaves* tl = nullptr;
        NdtP2dPose pose{};
        LoopMat4f final_transformation{};
        int evaluations = 0;
    };

    // computeAngleDerivatives: rows of d(Rx Ry Rz)/d(angle) and of the second derivatives (Magnusson eq. 6.19, 6.21)
    static void angle_derivatives(const double* p, NdtP2dPose& o) {
        double cx, cy, cz, sx, sy, sz;
        if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
        if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
        if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
        const double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                                {-sy * cz, sy * sz, cy}, {sx * cy * cz, -sx * cy * sz, sx * sy}, {-cx * cy * cz, cx * cy * sz, -cx * sy},
                                {-cy * sz, -cy * cz, 0.0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0.0}, {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0.0}};
        const double H[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                                 {cx * cy * cz, -cx * cy * sz, cx * sy}, {sx * cy * cz, -sx * cy * sz, sx * sy},
                                 {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0.0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0.0},
                                 {-cy * cz, cy * sz, -sy}, {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy},
                                 {sy * sz, sy * cz, 0.0}, {-sx * cy * sz, -sx * cy * cz, 0.0}, {cx * cy * sz, cx * cy * cz, 0.0},
                                 {-cy * cz, cy * sz, 0.0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0.0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0.0}};
        std::memcpy(o.j_ang, J, sizeof(J));
        std::memcpy(o.h_ang, H, sizeof(H));
    }

    // computeDerivatives / computeHessian on the device: score, gradient (6), Hessian (36, column-major)
    double ndt_eval(NdtRun& r, const LoopMat4f& T, const double* p, bool hessian, double* grad, double* hess) {
        angle_derivatives(p, r.pose);
        r.pose.T = T;
        const TargetLeaves& tl = *r.tl;
        NdtP2dTarget tg{tl.sparse ? nullptr : tl.d_leaf_row.p, tl.d_mean.p, tl.d_icov.p, tl.d_centroid.p, {tl.min_b[0], tl.min_b[1], tl.min_b[2]}, {tl.div_b[0], tl.div_b[1], tl.div_b[2]},
                        tl.inv, r.resolution * r.resolution, tl.sparse ? tl.d_leaf_hash.p : nullptr, tl.hash_mask};
        const int n = int(r.src->n), nb = (n + kLoopBlock - 1) / kLoopBlock;
        ++r.evaluations;
        const double* v = reduce(nb, hessian ? 44 : 8, [&](const LoopOut rows) {
            if (hessian) hipLaunchKernelGGL(ndt_p2d_kernel<true>, dim3(unsigned(nb)), dim3(kLoopBlock), 0, stream, r.src->x(), r.src->y(), r.src->z(), n, tg, r.pose, rows);
            else hipLaunchKernelGGL(ndt_p2d_kernel<false>, dim3(unsigned(nb)), dim3(kLoopBlock), 0, stream, r.src->x(), r.src->y(), r.src->z(), n, tg, r.pose, rows);
        });
        if (grad) for (int i = 0; i < 6; ++i) grad[i] = v[1 + i];
        if (hessian && hess) for (int i = 0; i < 36; ++i) hess[i] = v[7 + i];
        return v[0];
    }

    // JacobiSVD<Matrix6d>(A, ComputeFullU | ComputeFullV).solve(b)
    static void svd_solve6(const double* A, const double* b, double* x) {
        constexpr int N = 6;
        const double precision = 2.0 * std::numeric_limits<double>::epsilon(), tiny = std::numeric_limits<double>::min();
        double scale = 0.0;
        for (int i = 0; i < N * N; ++i) scale = std::max(scale, std::fabs(A[i]));
        if (scale == 0.0) scale = 1.0;
        double W[N * N], U[N * N], V[N * N];
        for (int i = 0; i < N * N; ++i) { W[i] = A[i] / scale; U[i] = V[i] = (i % (N + 1) == 0) ? 1.0 : 0.0; }
        double max_diag = 0.0;
        for (int i = 0; i < N; ++i) max_diag = std::max(max_diag, std::fabs(W[i + N * i]));
        auto jacobi = [&](double x_, double y_, double z_, double& c, double& s) {
            const double deno = 2.0 * std::fabs(y_);
            if (deno < tiny) { c = 1.0; s = 0.0; return; }
            const double tau = (x_ - z_) / deno, w = std::sqrt(tau * tau + 1.0);
            const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
            const double sign_t = t > 0.0 ? 1.0 : -1.0, nn = 1.0 / std::sqrt(t * t + 1.0);
            s = -sign_t * (y_ / std::fabs(y_)) * std::fabs(t) * nn;
            c = nn;
        };
        bool finished = false;
        for (int sweep = 0; !finished && sweep < 1000; ++sweep) {
            finished = true;
            for (int p = 1; p < N; ++p)
                for (int q = 0; q < p; ++q) {
                    const double threshold = std::max(tiny, precision * max_diag);
                    if (!(std::fabs(W[p + q * N]) > threshold || std::fabs(W[q + p * N]) > threshold)) continue;
                    finished = false;
                    const double m00 = W[p + p * N], m01 = W[p + q * N], m10 = W[q + p * N], m11 = W[q + q * N];
                    double r1c, r1s;
                    const double t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < tiny) { r1s = 0.0; r1c = 1.0; }
                    else { const double u = t / d, tmp = std::sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
                    const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
                    double jc, js;
                    jacobi(n00, n01, n11, jc, js);
                    const double lc = r1c * jc - r1s * (-js), ls = r1c * (-js) + r1s * jc;  // j_left = rot1 * j_right^T
                    for (int k = 0; k < N; ++k) {
                        const double xi = W[p + k * N], yi = W[q + k * N];
                        W[p + k * N] = lc * xi + ls * yi;
                        W[q + k * N] = -ls * xi + lc * yi;
                    }
                    for (int k = 0; k < N; ++k) {
                        const double xi = U[k + p * N], yi = U[k + q * N];
                        U[k + p * N] = lc * xi + ls * yi;
                        U[k + q * N] = -ls * xi + lc * yi;
                    }
                    for (int k = 0; k < N; ++k) {
                        double xi = W[k + p * N], yi = W[k + q * N];
                        W[k + p * N] = jc * xi - js * yi;
                        W[k + q * N] = js * xi + jc * yi;
                        xi = V[k + p * N]; yi = V[k + q * N];
                        V[k + p * N] = jc * xi - js * yi;
                        V[k + q * N] = js * xi + jc * yi;
                    }
                    max_diag = std::max(max_diag, std::max(std::fabs(W[p + p * N]), std::fabs(W[q + q * N])));
                }
        }
        double S[N], smax = 0.0;
        for (int i = 0; i < N; ++i) {
            const double a = W[i + i * N];
            S[i] = std::fabs(a) * scale;
            if (a < 0.0) for (int k = 0; k < N; ++k) U[k + i * N] = -U[k + i * N];
            smax = std::max(smax, S[i]);
        }
        const double thr = std::numeric_limits<double>::epsilon() * N * smax;
        for (int i = 0; i < N; ++i) x[i] = 0.0;
        for (int k = 0; k < N; ++k) {
            if (!(S[k] > thr)) continue;
            double ub = 0.0;
            for (int i = 0; i < N; ++i) ub += U[i + k * N] * b[i];
            const double c = ub / S[k];
            for (int i = 0; i < N; ++i) x[i] += V[i + k * N] * c;
        }
    }

    // More-Thuente pieces (ndt.hpp): psi, psi', interval update, trial value
    static double mt_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
    static double mt_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
    struct MtEnd { double a, f, g; };
    static bool mt_update(MtEnd& l, MtEnd& u, const MtEnd& t) {
        if (t.f > l.f) { u = t; return false; }
        const double s = t.g * (l.a - t.a);
        if (s > 0) { l = t; return false; }
        if (s < 0) { u = l; l = t; return false; }
        return true;
    }
    static double mt_cubic_min(const MtEnd& p, const MtEnd& q) {  // minimiser of the cubic through (p.a, p.f, p.g), (q.a, q.f, q.g)
        const double z = 3 * (q.f - p.f) / (q.a - p.a) - q.g - p.g, w = std::sqrt(z * z - q.g * p.g);
        return p.a + (q.a - p.a) * (w - p.g - z) / (q.g - p.g + 2 * w);
    }
    static double mt_trial(const MtEnd& l, const MtEnd& u, const MtEnd& t) {
        if (t.f > l.f) {
            const double a_c = mt_cubic_min(l, t), a_q = l.a - 0.5 * (l.a - t.a) * l.g / (l.g - (l.f - t.f) / (l.a - t.a));
            return std::fabs(a_c - l.a) < std::fabs(a_q - l.a) ? a_c : 0.5 * (a_q + a_c);
        }
        if (t.g * l.g < 0) {
            const double a_c = mt_cubic_min(l, t), a_s = l.a - (l.a - t.a) / (l.g - t.g) * l.g;
            return std::fabs(a_c - t.a) >= std::fabs(a_s - t.a) ? a_c : a_s;
        }
        if (std::fabs(t.g) <= std::fabs(l.g)) {
            const double a_c = mt_cubic_min(l, t), a_s = l.a - (l.a - t.a) / (l.g - t.g) * l.g;
            const double nxt = std::fabs(a_c - t.a) < std::fabs(a_s - t.a) ? a_c : a_s;
            return t.a > l.a ? std::min(t.a + 0.66 * (u.a - t.a), nxt) : std::max(t.a + 0.66 * (u.a - t.a), nxt);
        }
        return mt_cubic_min(u, t);
    }
    static double dot6(const double* a, const double* b) { double s = 0.0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }

    double step_length_mt(NdtRun& r, const double* x, double* dir, double step_init, double step_max, double step_min, double& score, double* grad, double* hess) {
        const double phi_0 = -score;
        double d_phi_0 = -dot6(grad, dir);
        if (d_phi_0 >= 0) {
            if (d_phi_0 == 0) return 0;
            d_phi_0 = -d_phi_0;
            for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
        }
        const double mu = 1.e-4, nu = 0.9;
        MtEnd lo{0.0, mt_psi(0.0, phi_0, phi_0, d_phi_0, mu), mt_dpsi(d_phi_0, d_phi_0, mu)}, up = lo;
        bool interval_converged = (step_max - step_min) < 0, open_interval = true;
        int trials = 0;
        double a_t = std::max(std::min(step_init, step_max), step_min), x_t[6];
        auto evaluate = [&](bool with_hessian) {
            for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
            r.final_transformation = pose_xyz(x_t);
            score = ndt_eval(r, r.final_transformation, x_t, with_hessian, grad, hess);
        };
        evaluate(true);
        double phi_t = -score, d_phi_t = -dot6(grad, dir);
        double psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
        while (!interval_converged && trials < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
            a_t = open_interval ? mt_trial(lo, up, MtEnd{a_t, psi_t, d_psi_t}) : mt_trial(lo, up, MtEnd{a_t, phi_t, d_phi_t});
            a_t = std::max(std::min(a_t, step_max), step_min);
            evaluate(false);
            phi_t = -score;
            d_phi_t = -dot6(grad, dir);
            psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu);
            d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
            if (open_interval && psi_t <= 0 && d_psi_t >= 0) {  // the interval is closed from here on: psi -> phi at both ends
                open_interval = false;
                lo.f = lo.f + phi_0 - mu * d_phi_0 * lo.a; lo.g = lo.g + mu * d_phi_0;
                up.f = up.f + phi_0 - mu * d_phi_0 * up.a; up.g = up.g + mu * d_phi_0;
            }
            interval_converged = open_interval ? mt_update(lo, up, MtEnd{a_t, psi_t, d_psi_t}) : mt_update(lo, up, MtEnd{a_t, phi_t, d_phi_t});
            ++trials;
        }
        if (trials) {  // computeHessian at the accepted point (score and gradient are current)
            double g_unused[6];
            ndt_eval(r, r.final_transformation, x_t, true, g_unused, hess);
        }
        return a_t;
    }

    LoopMat4f ndt_align(NdtRun& r, const LoopMat4f& guess, int& iterations, double& score_out) {
        const double c1 = 10.0 * (1.0 - r.outlier_ratio), c2 = r.outlier_ratio / std::pow(double(r.resolution), 3), d3 = -std::log(c2);
        r.pose.gauss_d1 = -std::log(c1 + c2) - d3;
        r.pose.gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / r.pose.gauss_d1);
        r.final_transformation = guess;
        iterations = 0;
        score_out = 0.0;
        if (!r.tl || r.tl->rows == 0 || r.src->n == 0) return r.final_transformation;
        float e[3];
        euler012(guess, e);
        double p[6] = {double(guess.m[12]), double(guess.m[13]), double(guess.m[14]), double(e[0]), double(e[1]), double(e[2])};
        double grad[6], hess[36], delta[6];
        double score = ndt_eval(r, r.final_transformation, p, true, grad, hess);
        for (bool converged = false; !converged;) {
            double neg[6];
            for (int i = 0; i < 6; ++i) neg[i] = -grad[i];
            svd_solve6(hess, neg, delta);
            double nrm = std::sqrt(dot6(delta, delta));
            if (nrm == 0 || nrm != nrm) break;
            for (int i = 0; i < 6; ++i) delta[i] /= nrm;
            nrm = step_length_mt(r, p, delta, nrm, r.step_size, r.transformation_epsilon / 2, score, grad, hess);
            for (int i = 0; i < 6; ++i) { delta[i] *= nrm; p[i] += delta[i]; }
            if (iterations > r.max_iterations || (iterations && std::fabs(nrm) < r.transformation_epsilon)) converged = true;
            ++iterations;
        }
        score_out = score / double(r.src->n);
        return r.final_transformation;
    }

    // ======================================================================================================================
    // GeneralizedIterativeClosestPoint
    // ======================================================================================================================
    struct GicpRun {
        DevCloud moved;           // `output`: the source transformed by the guess (float)
        CellGridImage tgt_grid;   // target cloud, ids = cloud indices, by_id copy for the cost kernel
        DevBuf<double> cov_src, cov_tgt, mahal;
        DevBuf<int> corr;
        size_t n_src = 0, n_tgt = 0;
        int evaluations = 0, inner_total = 0, n_corr = 0;
    };
    static void apply_state(LoopMat4f& t, const double* x) {  // gicp.hpp applyState: R = Rz(x5) Ry(x4) Rx(x3) on the left, translation added
        const LoopMat4f R = mul(mul(rot(2, float(x[5])), rot(1, float(x[4]))), rot(0, float(x[3])));
        LoopMat4f o = t;
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) o.m[i + 4 * j] = (R.m[i] * t.m[4 * j] + R.m[i + 4] * t.m[1 + 4 * j]) + R.m[i + 8] * t.m[2 + 4 * j];
        o.m[12] = t.m[12] + float(x[0]); o.m[13] = t.m[13] + float(x[1]); o.m[14] = t.m[14] + float(x[2]);
        t = o;
    }
    // OptimizationFunctorWithIndices: f (and g) at x, sums on the device
    void gicp_fdf(GicpRun& g, const double* x, double* f, double* grad) {
        ++g.evaluations;
        LoopMat4f T = ident();
        apply_state(T, x);
        const int n = int(g.n_src), nb = (n + kLoopBlock - 1) / kLoopBlock;
        const bool with_g = grad != nullptr;
        const double* v = reduce(nb, with_g ? 14 : 2, [&](const LoopOut rows) {
            if (with_g) hipLaunchKernelGGL(gicp_fdf_kernel<true>, dim3(unsigned(nb)), dim3(kLoopBlock), 0, stream, g.moved.x(), g.moved.y(), g.moved.z(), n, T,
                                           (const float4*)g.tgt_grid.d_by_id.p, (const int*)g.corr.p, (const double*)g.mahal.p, rows);
            else hipLaunchKernelGGL(gicp_fdf_kernel<false>, dim3(unsigned(nb)), dim3(kLoopBlock), 0, stream, g.moved.x(), g.moved.y(), g.moved.z(), n, T,
                                    (const float4*)g.tgt_grid.d_by_id.p, (const int*)g.corr.p, (const double*)g.mahal.p, rows);
        });
        const double m = v[with_g ? 13 : 1];
        if (f) *f = v[0] / m;
        if (!with_g) return;
        double Racc[9];
        for (int a = 0; a < 3; ++a) grad[a] = v[1 + a] * (2.0 / m);
        for (int q = 0; q < 9; ++q) Racc[q] = v[4 + q] * (2.0 / m);
        // computeRDerivative: derivatives of Rz(psi) Ry(theta) Rx(phi); g[3 + k] = trace(dR_k * Racc)
        const double cphi = std::cos(x[3]), sphi = std::sin(x[3]), cth = std::cos(x[4]), sth = std::sin(x[4]), cpsi = std::cos(x[5]), spsi = std::sin(x[5]);
        const double dphi[9] = {0.0, 0.0, 0.0,  // column 0
                                sphi * spsi + cphi * cpsi * sth, -cpsi * sphi + cphi * spsi * sth, cphi * cth,
                                cphi * spsi - cpsi * sphi * sth, -cphi * cpsi - sphi * spsi * sth, -cth * sphi};
        const double dth[9] = {-cpsi * sth, -spsi * sth, -cth, cpsi * cth * sphi, cth * sphi * spsi, -sphi * sth, cphi * cpsi * cth, cphi * cth * spsi, -cphi * sth};
        const double dpsi[9] = {-cth * spsi, cpsi * cth, 0.0, -cphi * cpsi - sphi * spsi * sth, -cphi * spsi + cpsi * sphi * sth, 0.0,
                                cpsi * sphi - cphi * spsi * sth, sphi * spsi + cphi * cpsi * sth, 0.0};
        auto tr = [&](const double* d) { double s = 0.0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s += d[j + 3 * i] * Racc[i + 3 * j]; return s; };
        grad[3] = tr(dphi); grad[4] = tr(dth); grad[5] = tr(dpsi);
    }

    // pcl/registration/bfgs.h = GSL vector_bfgs2: direction update + Fletcher's bracketing / sectioning line search with cached evaluations
    struct Bfgs {
        LoopMatcher* host;
        GicpRun* run;
        double x0[6], g0[6], p[6], xa[6], ga[6], gradient[6];
        double f = 0, fa = 0, dfa = 0, g0norm = 0, pnorm = 0, fp0 = 0, delta_f = 0, kx = 0, kf = 0, kg = 0, kdf = 0;
        static double nrm(const double* v) { return std::sqrt(dot6(v, v)); }
        void move_to(double a) { if (a == kx) return; for (int i = 0; i < 6; ++i) xa[i] = x0[i] + a * p[i]; kx = a; }
        double eval_f(double a) { if (a == kf) return fa; move_to(a); host->gicp_fdf(*run, xa, &fa, nullptr); kf = a; return fa; }
        double eval_df(double a) {
            if (a == kdf) return dfa;
            move_to(a);
            if (a != kg) { double ft; host->gicp_fdf(*run, xa, &ft, ga); kg = a; }
            dfa = dot6(ga, p); kdf = a;
            return dfa;
        }
        void eval_fdf(double a, double& fv, double& dfv) {
            if (a == kf && a == kdf) { fv = fa; dfv = dfa; return; }
            if (a == kf || a == kdf) { fv = eval_f(a); dfv = eval_df(a); return; }
            move_to(a);
            host->gicp_fdf(*run, xa, &fa, ga);
            kf = a; kg = a; dfa = dot6(ga, p); kdf = a;
            fv = fa; dfv = dfa;
        }
        void new_direction() { for (int i = 0; i < 6; ++i) { xa[i] = x0[i]; ga[i] = g0[i]; } kx = kf = kg = 0; dfa = dot6(ga, p); kdf = 0; }
        void start(const double* x) {
            delta_f = 0;
            host->gicp_fdf(*run, x, &f, gradient);
            for (int i = 0; i < 6; ++i) { x0[i] = x[i]; g0[i] = gradient[i]; }
            g0norm = nrm(g0);
            for (int i = 0; i < 6; ++i) p[i] = gradient[i] * (-1.0 / g0norm);
            pnorm = nrm(p);
            fp0 = -g0norm;
            for (int i = 0; i < 6; ++i) { xa[i] = x0[i]; ga[i] = g0[i]; }
            kx = 0; fa = f; kf = 0; kg = 0; dfa = dot6(ga, p); kdf = 0;
        }
        static double poly3(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
        static int quad_roots(double a, double b, double c, double& r0, double& r1) {  // gsl_poly_solve_quadratic
            const double disc = b * b - 4 * a * c;
            if (a == 0) { if (b == 0) return 0; r0 = -c / b; return 1; }
            if (disc > 0) {
                if (b == 0) { const double r = std::fabs(0.5 * std::sqrt(disc) / a); r0 = -r; r1 = r; }
                else { const double t = -0.5 * (b + (b > 0 ? 1.0 : -1.0) * std::sqrt(disc)), q1 = t / a, q2 = c / t; r0 = std::min(q1, q2); r1 = std::max(q1, q2); }
                return 2;
            }
            if (disc == 0) { r0 = r1 = -0.5 * b / a; return 2; }
            return 0;
        }
        static double min_quad(double f0, double d0, double f1, double zl, double zh) {
            const double k = f1 - f0 - d0, fl = f0 + zl * (d0 + zl * k), fh = f0 + zh * (d0 + zh * k), c = 2 * k;
            double zmin = zl, fmin = fl;
            if (fh < fmin) { zmin = zh; fmin = fh; }
            if (c > 0) { const double z = -d0 / c; if (z > zl && z < zh) { const double fz = f0 + z * (d0 + z * k); if (fz < fmin) { zmin = z; fmin = fz; } } }
            return zmin;
        }
        static double min_cubic(double f0, double d0, double f1, double d1, double zl, double zh) {
            const double c2 = 3 * (f1 - f0) - 2 * d0 - d1, c3 = d0 + d1 - 2 * (f1 - f0);
            double zmin = zl, fmin = poly3(f0, d0, c2, c3, zl), z0 = 0, z1 = 0;
            auto consider = [&](double z) { const double y = poly3(f0, d0, c2, c3, z); if (y < fmin) { zmin = z; fmin = y; } };
            consider(zh);
            const int n = quad_roots(3 * c3, 2 * c2, d0, z0, z1);
            if (n >= 1 && z0 > zl && z0 < zh) consider(z0);
            if (n == 2 && z1 > zl && z1 < zh) consider(z1);
            return zmin;
        }
        static double interpolate(double a, double fa_, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
            double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
            if (ymin > ymax) std::swap(ymin, ymax);
            const bool cubic_ok = !(fpb != fpb) && fpb != std::numeric_limits<double>::infinity();  // order 3
            const double y = cubic_ok ? min_cubic(fa_, fpa * (b - a), fb, fpb * (b - a), ymin, ymax) : min_quad(fa_, fpa * (b - a), fb, ymin, ymax);
            return a + y * (b - a);
        }
        // 0 = Success, 1 = NoProgress
        int line_search(double alpha1, double& alpha_new) {
            const double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, nan = std::numeric_limits<double>::quiet_NaN();
            double f0v, fp0v;
            eval_fdf(0.0, f0v, fp0v);
            double alpha = alpha1, alpha_prev = 0.0, falpha, fpalpha, falpha_prev = f0v, fpalpha_prev = fp0v;
            double a = 0.0, b = alpha, f_a = f0v, f_b = 0.0, fp_a = fp0v, fp_b = 0.0;
            int i = 0;
            while (i++ < 100) {  // bracketing
                falpha = eval_f(alpha);
                if (falpha > f0v + alpha * rho * fp0v || falpha >= falpha_prev) { a = alpha_prev; f_a = falpha_prev; fp_a = fpalpha_prev; b = alpha; f_b = falpha; fp_b = nan; break; }
                fpalpha = eval_df(alpha);
                if (std::fabs(fpalpha) <= -sigma * fp0v) { alpha_new = alpha; return 0; }
                if (fpalpha >= 0) { a = alpha; f_a = falpha; fp_a = fpalpha; b = alpha_prev; f_b = falpha_prev; fp_b = fpalpha_prev; break; }
                const double delta = alpha - alpha_prev;
                const double next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
                alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = next;
            }
            while (i++ < 100) {  // sectioning
                const double delta = b - a;
                alpha = interpolate(a, f_a, fp_a, b, f_b, fp_b, a + tau2 * delta, b - tau3 * delta);
                falpha = eval_f(alpha);
                if ((a - alpha) * fp_a <= std::numeric_limits<double>::epsilon()) return 1;
                if (falpha > f0v + rho * alpha * fp0v || falpha >= f_a) { b = alpha; f_b = falpha; fp_b = nan; }
                else {
                    fpalpha = eval_df(alpha);
                    if (std::fabs(fpalpha) <= -sigma * fp0v) { alpha_new = alpha; return 0; }
                    if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; f_b = f_a; fp_b = fp_a; }
                    a = alpha; f_a = falpha; fp_a = fpalpha;
                }
            }
            return 0;
        }
        int step(double* x) {
            const double f_before = f;
            if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) return 1;
            double alpha1 = 1.0, alpha = 0.0;
            if (delta_f < 0) alpha1 = std::min(1.0, 2.0 * std::max(-delta_f, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f_before)) / (-fp0));
            const int status = line_search(alpha1, alpha);
            if (status != 0) return status;
            eval_fdf(alpha, fa, dfa);
            for (int i = 0; i < 6; ++i) { x[i] = xa[i]; gradient[i] = ga[i]; }
            f = fa;
            delta_f = f - f_before;
            double dx0[6], dg0[6];
            for (int i = 0; i < 6; ++i) { dx0[i] = x[i] - x0[i]; dg0[i] = gradient[i] - g0[i]; }
            const double dxg = dot6(dx0, gradient), dgg = dot6(dg0, gradient), dxdg = dot6(dx0, dg0), dgn = nrm(dg0);
            double A = 0, B = 0;
            if (dxdg != 0) { B = dxg / dxdg; A = -(1.0 + dgn * dgn / dxdg) * B + dgg / dxdg; }
            for (int i = 0; i < 6; ++i) p[i] = (-A * dx0[i] + gradient[i]) + -B * dg0[i];
            for (int i = 0; i < 6; ++i) { g0[i] = gradient[i]; x0[i] = x[i]; }
            g0norm = nrm(g0);
            pnorm = nrm(p);
            const double dir = dot6(p, gradient) > 0 ? -1.0 : 1.0;
            for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
            pnorm = nrm(p);
            fp0 = dot6(p, g0);
            new_direction();
            return 0;
        }
    };

    // estimateRigidTransformationBFGS
    bool gicp_estimate(GicpRun& g, LoopMat4f& transformation) {
        if (g.n_corr < 4) return false;
        double x[6] = {double(transformation.m[12]), double(transformation.m[13]), double(transformation.m[14]),
                       std::atan2(double(transformation.m[2 + 4 * 1]), double(transformation.m[2 + 4 * 2])), std::asin(-double(transformation.m[2])),
                       std::atan2(double(transformation.m[1]), double(transformation.m[0]))};
        Bfgs b;
        b.host = this;
        b.run = &g;
        b.start(x);
        int inner = 0, result;
        do {
            ++inner;
            result = b.step(x);
            if (result) break;
            result = Bfgs::nrm(b.gradient) < 1e-2 ? 0 : -1;
        } while (result == -1 && inner < 20);
        g.inner_total += inner;
        if (!(result == 1 || result == 0 || inner == 20)) return false;
        transformation = ident();
        apply_state(transformation, x);
        return true;
    }

    // the ten filtered clouds of one Match, produced in the order the stages consume them
    struct FilterAhead {
        static constexpr int kN = 10;
        std::vector<PtI> out[kN];
        std::mutex mx;
        std::condition_variable cv;
        int ready = 0;
        bool stop = false;
        std::exception_ptr error;
        std::thread th;
        void start(const std::vector<PtI>& source, const std::vector<PtI>& target) {
            th = std::thread([this, &source, &target] {
                static const float resolution[4] = {10.0f, 5.0f, 3.0f, 2.0f};
                for (int k = 0; k < kN; ++k) {
                    {
                        std::lock_guard<std::mutex> lk(mx);
                        if (stop) return;
                    }
                    try {
                        const float l = k < 8 ? resolution[k / 2] * 0.2f : (k == 8 ? 0.5f : 0.4f);  // NDT stages: resolution * 0.2f; GICP: 0.5 / 0.4
                        out[k] = voxel_grid((k & 1) ? target : source, l);
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(mx);
                        error = std::current_exception();
                        ready = kN;
                        cv.notify_all();
                        return;
                    }
                    {
                        std::lock_guard<std::mutex> lk(mx);
                        ready = k + 1;
                    }
                    cv.notify_all();
                }
            });
        }
        std::vector<PtI> take(int k) {
            std::unique_lock<std::mutex> lk(mx);
            cv.wait(lk, [&] { return ready > k; });
            if (error) std::rethrow_exception(error);
            return std::move(out[k]);
        }
        ~FilterAhead() {
            if (th.joinable()) {
                { std::lock_guard<std::mutex> lk(mx); stop = true; }
                th.join();
            }
        }
    };

    // buffers that live as long as the matcher (one per device, fls_reg.hip): no allocation on the second Match
    DeviceGridBuilder builder;
    DevCloud src_dev, tgt_dev, src_own;
    TargetLeaves leaves;
    GicpRun g;
    CellGridImage src_grid;

    fls_status run(const std::vector<PtI>& source, const std::vector<PtI>& target, double* T, float* fitness) {
        std::memset(&st, 0, sizeof(st));
        *fitness = std::numeric_limits<float>::max();
        read_env();
        FLS_HIP(hipSetDevice(device));
        if (d_ticket.p) FLS_HIP(hipMemsetAsync(d_ticket.p, 0, kTicketWords * sizeof(unsigned), stream));  // (a Match that failed half-way must not poison the next)
        // FLS_HOST_TIMING=1: where the wall time of one Match goes (stderr)
        const bool timing = std::getenv("FLS_HOST_TIMING") && std::atoi(std::getenv("FLS_HOST_TIMING")) != 0;
        double t_filter = 0, t_leaves = 0, t_ndt = 0, t_gicp_setup = 0, t_gicp = 0, t_fit = 0;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        auto lap = [&](double& acc) { const double t = now(); acc += t - t0; t0 = t; };
        // ---- four NDT stages ----------------------------------------------------------------------------------------------
        static const float resolution[4] = {10.0f, 5.0f, 3.0f, 2.0f};
        // The ten VoxelGridCloud calls depend on the inputs only: with the exact host filter they run on a side thread (which
        // borrows the worker pool), one stage ahead of the optimiser that spins on the device's result block on this thread.
        FilterAhead ahead;
        if (!device_filter) ahead.start(source, target);
        auto filtered = [&](int stage, bool is_target, float leaf) -> std::vector<PtI> {
            if (!device_filter) return ahead.take(2 * stage + (is_target ? 1 : 0));
            return filter(is_target ? target : source, leaf);
        };
        for (int s = 0; s < 4; ++s) {
            const float r = resolution[s];
            const std::vector<PtI> src = filtered(s, false, r * 0.2f), tgt = filtered(s, true, r * 0.2f);
            lap(t_filter);
            NdtRun run;
            run.resolution = r;
            const bool have = build_target(tgt, r, leaves);
            src_dev.upload(src, stream);
            lap(t_leaves);
            run.src = &src_dev;
            run.tl = have ? &leaves : nullptr;
            int iters = 0;
            double score = 0.0;
            const LoopMat4f fin = ndt_align(run, from_d(T), iters, score);
            for (int i = 0; i < 16; ++i) T[i] = double(fin.m[i]);
            st.ndt_iterations[s] = iters; st.ndt_evaluations[s] = run.evaluations; st.ndt_source_points[s] = int(src.size());
            st.ndt_target_leaves[s] = have ? int(leaves.rows) : 0; st.ndt_score[s] = score;
            lap(t_ndt);
        }
        for (int i = 0; i < 16; ++i) st.T_after_ndt[i] = T[i];
        // ---- GICP ---------------------------------------------------------------------------------------------------------
        const std::vector<PtI> src = filtered(4, false, 0.5f), tgt = filtered(4, true, 0.4f);
        lap(t_filter);
        st.gicp_source_points = int(src.size());
        st.gicp_target_points = int(tgt.size());
        if (src.size() < 20 || tgt.size() < 20) return FLS_OK;  // computeCovariances refuses (k_correspondences_ > cloud size): no alignment
        g.n_src = src.size(); g.n_tgt = tgt.size();
        g.evaluations = g.inner_total = g.n_corr = 0;
        const LoopMat4f guess = from_d(T);
        // target grid (ids = cloud indices) + 20-NN covariances of both clouds, each in its own grid
        tgt_dev.upload(tgt, stream);
        src_own.upload(src, stream);
        const float cell = 1.5f;  // 20 neighbours of a 0.4-0.5 m down-sampled cloud lie within ~1.2 m on a surface: the 27 cells certify most queries
        if (!builder.run(g.tgt_grid, tgt_dev.x(), tgt_dev.y(), tgt_dev.z(), tgt.size(), cell, 1, true, stream)) {
            const fls_status rc = g.tgt_grid.build(tgt, cell, stream, 1, true);
            if (rc != FLS_OK) return rc;
        }
        if (!builder.run(src_grid, src_own.x(), src_own.y(), src_own.z(), src.size(), cell, 1, false, stream)) {
            const fls_status rc = src_grid.build(src, cell, stream, 1, false);
            if (rc != FLS_OK) return rc;
        }
        g.cov_src.reserve(src.size() * 9); g.cov_tgt.reserve(tgt.size() * 9); g.mahal.reserve(src.size() * 9); g.corr.reserve(src.size());
        hipLaunchKernelGGL(gicp_cov_kernel, dim3(unsigned((tgt.size() + kCovWaves - 1) / kCovWaves)), dim3(64 * kCovWaves), 0, stream, tgt_dev.x(), tgt_dev.y(), tgt_dev.z(), int(tgt.size()),
                           cell_dev(g.tgt_grid), 0.001, g.cov_tgt.p);
        hipLaunchKernelGGL(gicp_cov_kernel, dim3(unsigned((src.size() + kCovWaves - 1) / kCovWaves)), dim3(64 * kCovWaves), 0, stream, src_own.x(), src_own.y(), src_own.z(), int(src.size()),
                           cell_dev(src_grid), 0.001, g.cov_src.p);
        FLS_HIP(hipGetLastError());
        // `output` = the source moved by the guess
        {
            std::vector<PtI> moved(src.size());
            for (size_t i = 0; i < src.size(); ++i) { float o[3]; loop_xform(guess, src[i].x, src[i].y, src[i].z, o); moved[i] = PtI{o[0], o[1], o[2], src[i].i}; }
            g.moved.upload(moved, stream);
        }
        if (timing) FLS_HIP(hipStreamSynchronize(stream));
        lap(t_gicp_setup);
        LoopMat4f transformation = ident(), previous = ident();
        const double corr_dist = 2.0, dist_threshold = corr_dist * corr_dist, rotation_epsilon = 2e-3, transformation_epsilon = 5e-4;
        const CellGridDev cg_tgt = cell_dev(g.tgt_grid);
        int nr = 0;
        for (bool converged = false; !converged;) {
            GicpRot rot;
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) {
                    double sum = 0.0;
                    for (int k = 0; k < 4; ++k) sum += double(transformation.m[i + 4 * k]) * double(guess.m[k + 4 * j]);
                    rot.R[i + 3 * j] = sum;
                }
            const int n = int(src.size());
            hipLaunchKernelGGL(gicp_corr_kernel, dim3(unsigned((n + 63) / 64)), dim3(64), 0, stream, g.moved.x(), g.moved.y(), g.moved.z(), n, transformation, cg_tgt,
                               float(dist_threshold), dist_threshold, rot, (const double*)g.cov_src.p, (const double*)g.cov_tgt.p, g.corr.p, g.mahal.p);
            FLS_HIP(hipGetLastError());
            {   // number of correspondences (the cost kernel's count column at the current pose; also warms nothing else)
                double f0;
                const double x0[6] = {0, 0, 0, 0, 0, 0};
                const int before = g.evaluations;
                gicp_fdf(g, x0, &f0, nullptr);
                g.evaluations = before;
                g.n_corr = int(mail_host->v[1]);
            }
            previous = transformation;
            if (!gicp_estimate(g, transformation)) { st.gicp_failed = 1; break; }
            double delta = 0.0;
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 4; ++l) {
                    const double ratio = (k < 3 && l < 3) ? 1.0 / rotation_epsilon : 1.0 / transformation_epsilon;
                    delta = std::max(delta, ratio * std::fabs(double(previous.m[k + 4 * l]) - double(transformation.m[k + 4 * l])));
                }
            ++nr;
            if (debug) std::fprintf(stderr, "[fls loop] gicp outer %d: corr %d inner_total %d evals %d delta %.9g t = %.9g %.9g %.9g\n", nr, g.n_corr, g.inner_total, g.evaluations, delta,
                                    double(transformation.m[12]), double(transformation.m[13]), double(transformation.m[14]));
            if (nr >= 30 || delta < 1) { converged = true; previous = transformation; }
        }
        lap(t_gicp);
        const LoopMat4f fin = mul(previous, guess);
        for (int i = 0; i < 16; ++i) T[i] = double(fin.m[i]);
        st.gicp_iterations = nr; st.gicp_inner_iterations = g.inner_total; st.gicp_evaluations = g.evaluations; st.gicp_correspondences = g.n_corr;
        // ---- getFitnessScore ------------------------------------------------------------------------------------------------
        {
            const int n = int(src.size()), nb = (n + kLoopBlock - 1) / kLoopBlock;
            const double* v = reduce(nb, 2, [&](const LoopOut rows) {
                hipLaunchKernelGGL(loop_fitness_kernel, dim3(unsigned(nb)), dim3(kLoopBlock), 0, stream, src_own.x(), src_own.y(), src_own.z(), n, fin, cg_tgt, rows);
            });
            if (v[1] > 0) *fitness = float(v[0] / v[1]);
        }
        FLS_HIP(hipStreamSynchronize(stream));
        lap(t_fit);
        if (timing)
            std::fprintf(stderr, "[fls loop] ms: 10 filters %.2f, leaf Gaussians + uploads %.2f, NDT align (%d evaluations) %.2f, GICP grids + covariances %.2f, GICP loop (%d evaluations) %.2f, fitness %.2f\n",
                         t_filter, t_leaves, st.ndt_evaluations[0] + st.ndt_evaluations[1] + st.ndt_evaluations[2] + st.ndt_evaluations[3], t_ndt, t_gicp_setup, st.gicp_evaluations, t_gicp, t_fit);
        return FLS_OK;
    }
};

}  // namespace fls
