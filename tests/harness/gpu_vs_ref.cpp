// ============================================================================
// tests/harness/gpu_vs_ref.cpp  --  TEST INFRASTRUCTURE ONLY (VERDICT r2 "missing #5", "next #1b").
//
// ONE C++ program that holds two std::shared_ptr<RegistrationInterface>:
//     * the REFERENCE'S OWN class (LoamPointToPlaneIVOX<double> / IcpOptimized<double> / IncrementalNDT /
//       LoamFull<double> / LoamPointToPlaneKdtree<double>), compiled verbatim from /root/reference against the
//       include-shadow shim of oracle/ref_shim, constructed as FrontEnd::InitMatcher / Localization::InitMatcher do
//       (src/slam/frontend.cpp:30-88, src/slam/localization.cpp:43-92), and
//     * HipRegistration (include/fls_hip_registration.h) -- compiled HERE against the reference's REAL
//       registration/registration_interface.h, common/data_type.h and lidar/pointcloud_cluster.h,
// and drives both through the same call sequence the pipeline issues (src/slam/frontend.cpp:125-140: the first
// cloud(s) through AddCloudToLocalMap in the world frame; :208: Match(cluster, predicted pose) once per scan, Match
// itself growing the map; localization.cpp:135-138: AddCloudToLocalMap + Match + GetFitnessScore(2.0)).
// The two matchers never see each other's results: each follows its own pose chain (T_prev * guess_step).
//
// One scenario per process (the reference keeps function-static state).  Built by
// oracle/ref_shim/Makefile into oracle/_ref/gpu_vs_ref (git-ignored, travels to the GPU box);
// driven by tests/test_gpu_vs_ref.py, which writes the scenario file and checks the per-frame records.
//
//   usage: gpu_vs_ref <scenario.bin> <out.txt> [--ref-only]   |   gpu_vs_ref --compile-check
// ============================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <execution>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include <Eigen/Dense>
#include <glog/logging.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/common/transforms.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/kdtree/kdtree_flann.h>

#define private public  // only for the iVox LRU-capacity test hook (the reference hard-codes 1e6, ivox_map.h:36)
#include "registration/loam_point_to_plane_ivox.h"
#undef private
#include "registration/icp_optimized.h"
#include "registration/incremental_ndt.h"
#include "registration/loam_full_kdtree.h"
#include "registration/loam_point_to_plane_kdtree.h"

#include "fls_hip_registration.h"  // the product's adapter, against the REAL reference headers

namespace {

struct Frame {
    PCLPointCloudXYZI scan, corner;
    double step[16];  // column-major; guess = T_prev * step, or = step when `absolute`
    int absolute = 0;
};
struct Scenario {
    int kind = 0, loc = 0;
    long long ivox_capacity = 0;
    fls_params p{};
    std::vector<PCLPointCloudXYZI> init;
    std::vector<Frame> frames;
};

bool read_exact(FILE* f, void* dst, size_t n) { return n == 0 || std::fread(dst, 1, n, f) == n; }

bool read_cloud(FILE* f, PCLPointCloudXYZI& c) {
    std::uint64_t n = 0;
    if (!read_exact(f, &n, 8)) return false;
    std::vector<float> buf(size_t(n) * 4);
    if (!read_exact(f, buf.data(), buf.size() * 4)) return false;
    c.points.resize(size_t(n));
    for (size_t i = 0; i < size_t(n); ++i) {
        PCLPointXYZI q;
        q.x = buf[4 * i]; q.y = buf[4 * i + 1]; q.z = buf[4 * i + 2]; q.intensity = buf[4 * i + 3];
        c.points[i] = q;
    }
    c.width = std::uint32_t(n); c.height = 1;
    return true;
}

bool load(const char* path, Scenario& s) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    std::int32_t hdr[4];
    std::uint32_t psize = 0;
    bool ok = read_exact(f, magic, 8) && std::memcmp(magic, "FLSSCN1", 8) == 0 && read_exact(f, hdr, sizeof(hdr)) && read_exact(f, &s.ivox_capacity, 8) &&
              read_exact(f, &psize, 4) && psize == sizeof(fls_params) && read_exact(f, &s.p, sizeof(fls_params));
    if (ok) {
        s.kind = hdr[0]; s.loc = hdr[1];
        s.init.resize(size_t(hdr[2]));
        for (auto& c : s.init) ok = ok && read_cloud(f, c);
        s.frames.resize(size_t(hdr[3]));
        for (auto& fr : s.frames) {
            ok = ok && read_cloud(f, fr.scan) && read_cloud(f, fr.corner) && read_exact(f, fr.step, sizeof(fr.step));
            std::int32_t a = 0;
            ok = ok && read_exact(f, &a, 4);
            fr.absolute = a;
        }
    }
    std::fclose(f);
    return ok;
}

// the reference class behind the interface, with the constructor argument lists of frontend.cpp:30-88 / localization.cpp:43-92
std::shared_ptr<RegistrationInterface> make_reference(const Scenario& s) {
    const fls_params& p = s.p;
    const bool loc = s.loc != 0;
    switch (s.kind) {
        case FLS_P2PLANE_IVOX: {
            auto q = std::make_shared<LoamPointToPlaneIVOX<double>>(p.point_to_planar_thres, p.position_converge_thres, p.rotation_converge_thres,
                                                                    size_t(p.max_iterations), loc);
            if (s.ivox_capacity > 0) q->ivox_map_ptr_->options_.capacity_ = size_t(s.ivox_capacity);
            return q;
        }
        case FLS_ICP_OPTIMIZED:
            return std::make_shared<IcpOptimized<double>>(p.max_iterations, p.local_map_size, p.map_cloud_filter_size, p.source_cloud_filter_size,
                                                          p.point_search_thres, p.position_converge_thres, p.rotation_converge_thres,
                                                          p.rot_thre_add_cloud, p.dist_thre_add_cloud, loc);
        case FLS_INCREMENTAL_NDT:
            return std::make_shared<IncrementalNDT>(p.ndt_voxel_size, p.ndt_res_outlier_threshold, p.source_cloud_filter_size, p.rotation_converge_thres,
                                                    p.position_converge_thres, p.ndt_min_points_in_voxel, p.ndt_max_points_in_voxel,
                                                    p.ndt_min_effective_pts, p.ndt_capacity, int(p.max_iterations), loc);
        case FLS_LOAM_FULL:
            return std::make_shared<LoamFull<double>>(p.point_to_planar_thres, p.point_search_thres, p.line_ratio_thres, p.position_converge_thres,
                                                      p.rotation_converge_thres, p.dist_thre_add_cloud, p.rot_thre_add_cloud, size_t(p.local_corner_size),
                                                      size_t(p.local_planar_size), p.corner_voxel_filter_size, p.planar_voxel_filter_size,
                                                      int(p.max_iterations));
        case FLS_P2PLANE_KDTREE:
            return std::make_shared<LoamPointToPlaneKdtree<double>>(p.point_to_planar_thres, p.position_converge_thres, p.rotation_converge_thres,
                                                                    p.rot_thre_add_cloud, p.dist_thre_add_cloud, p.local_map_size, p.map_cloud_filter_size,
                                                                    size_t(p.max_iterations), loc);
        default: return nullptr;
    }
}

// the product behind the same interface, through the factories that carry the reference constructors' argument lists
std::shared_ptr<RegistrationInterface> make_hip(const Scenario& s) {
    const fls_params& p = s.p;
    const bool loc = s.loc != 0;
    switch (s.kind) {
        case FLS_P2PLANE_IVOX: return HipRegistration::PointToPlaneIVOX(p.point_to_planar_thres, p.position_converge_thres, p.rotation_converge_thres, size_t(p.max_iterations), loc);
        case FLS_ICP_OPTIMIZED:
            return HipRegistration::IcpOptimized(p.max_iterations, p.local_map_size, p.map_cloud_filter_size, p.source_cloud_filter_size, p.point_search_thres,
                                                 p.position_converge_thres, p.rotation_converge_thres, p.rot_thre_add_cloud, p.dist_thre_add_cloud, loc);
        case FLS_INCREMENTAL_NDT:
            return HipRegistration::IncrementalNDT(p.ndt_voxel_size, p.ndt_res_outlier_threshold, p.source_cloud_filter_size, p.rotation_converge_thres,
                                                   p.position_converge_thres, p.ndt_min_points_in_voxel, p.ndt_max_points_in_voxel, p.ndt_min_effective_pts,
                                                   p.ndt_capacity, int(p.max_iterations), loc);
        case FLS_LOAM_FULL:
            return HipRegistration::LoamFull(p.point_to_planar_thres, p.point_search_thres, p.line_ratio_thres, p.position_converge_thres,
                                             p.rotation_converge_thres, p.dist_thre_add_cloud, p.rot_thre_add_cloud, size_t(p.local_corner_size),
                                             size_t(p.local_planar_size), p.corner_voxel_filter_size, p.planar_voxel_filter_size, int(p.max_iterations));
        case FLS_P2PLANE_KDTREE:
            return HipRegistration::PointToPlaneKdTree(p.point_to_planar_thres, p.position_converge_thres, p.rotation_converge_thres, p.rot_thre_add_cloud,
                                                       p.dist_thre_add_cloud, size_t(p.local_map_size), p.map_cloud_filter_size, size_t(p.max_iterations), loc);
        default: return nullptr;
    }
}

struct Record { int ok = 0; double T[16]{}; float fitness = 0.f; double ms = 0.0; };

// the pipeline's call sequence against ANY RegistrationInterface (this function does not know which one it drives)
std::vector<Record> replay(const std::shared_ptr<RegistrationInterface>& matcher, const Scenario& s) {
    if (s.kind == FLS_LOAM_FULL) matcher->AddCloudToLocalMap({s.init[0], s.init[1]});  // frontend.cpp:126-130 {planar, corner}
    else matcher->AddCloudToLocalMap({s.init[0]});                                      // :131-139, localization.cpp:135
    std::vector<Record> out;
    Mat4d T_prev = Mat4d::Identity();
    for (const Frame& fr : s.frames) {
        auto cluster = std::make_shared<PointcloudCluster>();
        if (s.kind == FLS_ICP_OPTIMIZED || s.kind == FLS_INCREMENTAL_NDT) cluster->ordered_cloud_ = fr.scan;
        else cluster->planar_cloud_ = fr.scan;
        if (s.kind == FLS_LOAM_FULL) cluster->corner_cloud_ = fr.corner;
        Mat4d step;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) step(i, j) = fr.step[i + 4 * j];
        Mat4d match_pose = step;
        if (!fr.absolute) match_pose = T_prev * step;
        Record r;
        const auto t0 = std::chrono::steady_clock::now();
        r.ok = matcher->Match(cluster, match_pose) ? 1 : 0;  // frontend.cpp:208
        r.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) r.T[i + 4 * j] = match_pose(i, j);
        if (s.loc) r.fitness = matcher->GetFitnessScore(2.0f);  // localization.cpp:138
        out.push_back(r);
        T_prev = match_pose;  // T is written even when Match returns false (tests/refpin.py follows the same chain)
    }
    return out;
}

void pose_diff(const double* A, const double* B, double& dt, double& dr) {
    dt = std::sqrt((A[12] - B[12]) * (A[12] - B[12]) + (A[13] - B[13]) * (A[13] - B[13]) + (A[14] - B[14]) * (A[14] - B[14]));
    double tr = 0.0;  // trace(Ra^T Rb)
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) tr += A[i + 4 * j] * B[i + 4 * j];
    // angle from the skew part (accurate for tiny angles, where acos((tr-1)/2) loses everything below 1e-8)
    double M[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0.0; for (int k = 0; k < 3; ++k) v += A[k + 4 * i] * B[k + 4 * j]; M[i + 3 * j] = v; }
    const double sx = M[5] - M[7], sy = M[6] - M[2], sz = M[1] - M[3];
    const double s = 0.5 * std::sqrt(sx * sx + sy * sy + sz * sz), c = 0.5 * (tr - 1.0);
    dr = std::atan2(s, c);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && std::strcmp(argv[1], "--compile-check") == 0) {
        std::printf("gpu_vs_ref: HipRegistration and the reference classes compiled against the reference's real headers\n");
        return 0;
    }
    if (argc < 3) { std::fprintf(stderr, "usage: gpu_vs_ref <scenario.bin> <out.txt>\n"); return 2; }
    Scenario s;
    if (!load(argv[1], s)) { std::fprintf(stderr, "gpu_vs_ref: cannot read scenario %s\n", argv[1]); return 2; }
    const bool ref_only = argc >= 4 && std::strcmp(argv[3], "--ref-only") == 0;  // CPU check of the scenario file + the reference leg
    if (ref_only) {
        const std::vector<Record> a = replay(make_reference(s), s);
        for (size_t k = 0; k < a.size(); ++k) std::printf("frame %zu ref_ok %d t %.9f %.9f %.9f ms %.2f\n", k, a[k].ok, a[k].T[12], a[k].T[13], a[k].T[14], a[k].ms);
        return 0;
    }
    if (fls_device_count() < 1) { std::fprintf(stderr, "gpu_vs_ref: no gfx950 device\n"); return 3; }
    if (s.ivox_capacity > 0) {
        const std::string cap = std::to_string(s.ivox_capacity);
        setenv("FLS_IVOX_CAPACITY", cap.c_str(), 1);  // the product's test hook for the same hard-coded constant
    }
    std::shared_ptr<RegistrationInterface> reference = make_reference(s), hip = make_hip(s);
    if (!reference || !hip) return 2;
    const std::vector<Record> a = replay(reference, s), b = replay(hip, s);
    FILE* o = std::fopen(argv[2], "w");
    if (!o) return 2;
    double worst_dt = 0.0, worst_dr = 0.0;
    int mismatched_returns = 0;
    for (size_t k = 0; k < a.size(); ++k) {
        double dt, dr;
        pose_diff(a[k].T, b[k].T, dt, dr);
        worst_dt = std::max(worst_dt, dt); worst_dr = std::max(worst_dr, dr);
        if (a[k].ok != b[k].ok) ++mismatched_returns;
        std::fprintf(o, "frame %zu ref_ok %d hip_ok %d dt %.17g dr %.17g ref_fitness %.9g hip_fitness %.9g ref_ms %.3f hip_ms %.3f\n", k, a[k].ok, b[k].ok, dt, dr,
                     double(a[k].fitness), double(b[k].fitness), a[k].ms, b[k].ms);
        std::fprintf(o, "ref_T");
        for (int q = 0; q < 16; ++q) std::fprintf(o, " %.17g", a[k].T[q]);
        std::fprintf(o, "\nhip_T");
        for (int q = 0; q < 16; ++q) std::fprintf(o, " %.17g", b[k].T[q]);
        std::fprintf(o, "\n");
    }
    std::fprintf(o, "summary frames %zu mismatched_returns %d worst_dt %.17g worst_dr %.17g\n", a.size(), mismatched_returns, worst_dt, worst_dr);
    std::fclose(o);
    std::printf("gpu_vs_ref: %zu frames, mismatched returns %d, worst |dt| %.3e m, worst |dR| %.3e rad\n", a.size(), mismatched_returns, worst_dt, worst_dr);
    return (mismatched_returns == 0 && worst_dt <= 1e-4 && worst_dr <= 1e-4) ? 0 : 1;
}
