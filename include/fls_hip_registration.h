// fls_hip_registration.h -- header-only adapter: the reference's RegistrationInterface implemented on
// top of libfls_reg.so (include/fls_reg.h).  Drop this header and the shared library into the
// funny_lidar_slam tree and select it with one of the *_HIP mode strings; nothing
// else in the ROS pipeline changes.
//
//   class RegistrationInterface                     include/registration/registration_interface.h:11-20
//   construction sites                              src/slam/frontend.cpp:30-88, src/slam/localization.cpp:43-92
//   call sites      Match                           src/slam/frontend.cpp:208, src/slam/localization.cpp:136,247
//                   AddCloudToLocalMap              src/slam/frontend.cpp:130,134,139, src/slam/localization.cpp:135,222
//                   GetFitnessScore                 src/slam/localization.cpp:138
//
// It needs only what the reference's own matchers already include: common/data_type.h (PCLPointCloudXYZI,
// Mat4d), lidar/pointcloud_cluster.h (PointcloudCluster) and registration/registration_interface.h.
// pcl::PointXYZI is 8 floats (x y z pad | intensity pad pad pad), so clouds are handed over in place
// with stride_floats = 8; Eigen::Matrix4d::data() is the 16-double column-major pose the ABI expects.
#ifndef FLS_HIP_REGISTRATION_H
#define FLS_HIP_REGISTRATION_H

#include "registration/registration_interface.h"
#include "fls_reg.h"

#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <limits>
#include <string>

// new mode strings, to be listed next to include/common/constant_variable.h:21-25
static const std::string kPointToPlane_IVOX_HIP = "PointToPlane_IVOX_HIP";
static const std::string kPointToPlane_KdTree_HIP = "PointToPlane_KdTree_HIP";
static const std::string kLoamFull_KdTree_HIP = "LoamFull_KdTree_HIP";
static const std::string kIcpOptimized_HIP = "IcpOptimized_HIP";
static const std::string kIncrementalNDT_HIP = "IncrementalNDT_HIP";

class HipRegistration final : public RegistrationInterface {
public:
    // ---- factories with the reference constructors' argument lists -------------------------------------
    // LoamPointToPlaneIVOX<double>(...)                      loam_point_to_plane_ivox.h:37-41
    static std::shared_ptr<HipRegistration> PointToPlaneIVOX(double point_to_planar_thres, double position_converge_thres,
                                                             double rotation_converge_thres, size_t opti_iter_num = 30u,
                                                             bool is_localization_mode = false, int device = 0) {
        fls_params p = Blank();
        p.point_to_planar_thres = point_to_planar_thres;
        p.position_converge_thres = position_converge_thres;
        p.rotation_converge_thres = rotation_converge_thres;
        p.max_iterations = static_cast<uint32_t>(opti_iter_num);
        p.is_localization_mode = is_localization_mode;
        return std::make_shared<HipRegistration>(FLS_P2PLANE_IVOX, p, device);
    }
    // IcpOptimized<double>(...)                              icp_optimized.h:23-26
    static std::shared_ptr<HipRegistration> IcpOptimized(unsigned int max_iterations, unsigned int local_map_size,
                                                         float map_cloud_filter_size, float source_cloud_filter_size,
                                                         double max_correspond_distance, double position_converge_thres,
                                                         double rotation_converge_thres, double rot_thre_add_cloud,
                                                         double dist_thre_add_cloud, bool is_localization_mode = false, int device = 0) {
        fls_params p = Blank();
        p.max_iterations = max_iterations;
        p.local_map_size = local_map_size;
        p.map_cloud_filter_size = map_cloud_filter_size;
        p.source_cloud_filter_size = source_cloud_filter_size;
        p.point_search_thres = max_correspond_distance;
        p.position_converge_thres = position_converge_thres;
        p.rotation_converge_thres = rotation_converge_thres;
        p.rot_thre_add_cloud = rot_thre_add_cloud;
        p.dist_thre_add_cloud = dist_thre_add_cloud;
        p.is_localization_mode = is_localization_mode;
        return std::make_shared<HipRegistration>(FLS_ICP_OPTIMIZED, p, device);
    }
    // IncrementalNDT(...)                                    incremental_ndt.h:22-26
    static std::shared_ptr<HipRegistration> IncrementalNDT(double voxel_size, double res_outlier_threshold,
                                                           float source_cloud_filter_size, double rotation_converge_thres,
                                                           double position_converge_thres, int min_points_in_voxel,
                                                           int max_points_in_voxel, int min_effective_pts, int capacity,
                                                           int max_iteration, bool is_localization_mode = false, int device = 0) {
        fls_params p = Blank();
        p.ndt_voxel_size = voxel_size;
        p.ndt_res_outlier_threshold = res_outlier_threshold;
        p.source_cloud_filter_size = source_cloud_filter_size;
        p.rotation_converge_thres = rotation_converge_thres;
        p.position_converge_thres = position_converge_thres;
        p.ndt_min_points_in_voxel = min_points_in_voxel;
        p.ndt_max_points_in_voxel = max_points_in_voxel;
        p.ndt_min_effective_pts = min_effective_pts;
        p.ndt_capacity = capacity;
        p.max_iterations = static_cast<uint32_t>(max_iteration);
        p.is_localization_mode = is_localization_mode;
        return std::make_shared<HipRegistration>(FLS_INCREMENTAL_NDT, p, device);
    }
    // LoamFull<double>(...)                                  loam_full_kdtree.h:31-36
    static std::shared_ptr<HipRegistration> LoamFull(double point_to_planar_thres, double point_search_thres, double line_ratio_thres,
                                                     double position_converge_thres, double rotation_converge_thres,
                                                     double dist_thre_add_cloud, double rot_thre_add_cloud, size_t local_corner_size,
                                                     size_t local_planar_size, float corner_voxel_filter_size,
                                                     float planar_voxel_filter_size, int max_iteration, int device = 0) {
        fls_params p = Blank();
        p.point_to_planar_thres = point_to_planar_thres;
        p.point_search_thres = point_search_thres;
        p.line_ratio_thres = line_ratio_thres;
        p.position_converge_thres = position_converge_thres;
        p.rotation_converge_thres = rotation_converge_thres;
        p.dist_thre_add_cloud = dist_thre_add_cloud;
        p.rot_thre_add_cloud = rot_thre_add_cloud;
        p.local_corner_size = static_cast<uint32_t>(local_corner_size);
        p.local_planar_size = static_cast<uint32_t>(local_planar_size);
        p.corner_voxel_filter_size = corner_voxel_filter_size;
        p.planar_voxel_filter_size = planar_voxel_filter_size;
        p.max_iterations = static_cast<uint32_t>(max_iteration);
        return std::make_shared<HipRegistration>(FLS_LOAM_FULL, p, device);
    }
    // LoamPointToPlaneKdtree<double>(...)                    loam_point_to_plane_kdtree.h:32-37
    static std::shared_ptr<HipRegistration> PointToPlaneKdTree(double point_to_planar_thres, double position_converge_thres,
                                                               double rotation_converge_thres, double rot_thre_add_cloud,
                                                               double dist_thre_add_cloud, size_t local_map_size,
                                                               float map_cloud_filter_size, size_t opti_iter_num = 30u,
                                                               bool is_localization_mode = false, int device = 0) {
        fls_params p = Blank();
        p.point_to_planar_thres = point_to_planar_thres;
        p.position_converge_thres = position_converge_thres;
        p.rotation_converge_thres = rotation_converge_thres;
        p.rot_thre_add_cloud = rot_thre_add_cloud;
        p.dist_thre_add_cloud = dist_thre_add_cloud;
        p.local_map_size = static_cast<uint32_t>(local_map_size);
        p.map_cloud_filter_size = map_cloud_filter_size;
        p.max_iterations = static_cast<uint32_t>(opti_iter_num);
        p.is_localization_mode = is_localization_mode;
        return std::make_shared<HipRegistration>(FLS_P2PLANE_KDTREE, p, device);
    }

    HipRegistration(fls_kind kind, const fls_params& params, int device = 0) : kind_(kind) {
        const fls_status rc = fls_create(kind, &params, device, &handle_);
        if (rc != FLS_OK) {
            // the reference's constructors CHECK-abort on bad parameters (icp_optimized.h:33-41); do the same
            std::fprintf(stderr, "HipRegistration: fls_create failed: %s\n", fls_status_string(rc));
            std::abort();
        }
    }
    ~HipRegistration() override { fls_destroy(handle_); }
    HipRegistration(const HipRegistration&) = delete;
    HipRegistration& operator=(const HipRegistration&) = delete;

    bool Match(const PointcloudClusterPtr& source_cloud_cluster, Mat4d& T) override {
        const bool ordered = (kind_ == FLS_ICP_OPTIMIZED || kind_ == FLS_INCREMENTAL_NDT);  // icp_optimized.h:57, incremental_ndt.h:232
        const PCLPointCloudXYZI& s0 = ordered ? source_cloud_cluster->ordered_cloud_ : source_cloud_cluster->planar_cloud_;
        const PCLPointCloudXYZI* s1 = (kind_ == FLS_LOAM_FULL) ? &source_cloud_cluster->corner_cloud_ : nullptr;
        const fls_status rc = fls_match(handle_, Data(s0), s0.size(), s1 ? Data(*s1) : nullptr, s1 ? s1->size() : 0, kStride,
                                        T.data(), /*update_map=*/1, &stats_);
        if (rc < 0) std::fprintf(stderr, "HipRegistration::Match: %s\n", fls_status_string(rc));
        return rc == FLS_OK;  // errors and FLS_NOT_CONVERGED both mean "drop this frame" (frontend.cpp:208-210)
    }

    void AddCloudToLocalMap(const std::initializer_list<PCLPointCloudXYZI>& cloud_list) override {
        const PCLPointCloudXYZI* c0 = cloud_list.begin();
        const PCLPointCloudXYZI* c1 = cloud_list.size() > 1 ? cloud_list.begin() + 1 : nullptr;
        const fls_status rc = fls_add_cloud_to_local_map(handle_, Data(*c0), c0->size(), c1 ? Data(*c1) : nullptr, c1 ? c1->size() : 0, kStride);
        if (rc != FLS_OK) {
            std::fprintf(stderr, "HipRegistration::AddCloudToLocalMap: %s\n", fls_status_string(rc));
            std::abort();  // CHECK_EQ(cloud_list.size(), ...) in the reference
        }
    }

    [[nodiscard]] float GetFitnessScore(float max_range) const override {
        float score = std::numeric_limits<float>::max();  // FloatNaN
        const fls_status rc = fls_get_fitness_score(handle_, max_range, &score);
        if (rc != FLS_OK) return std::numeric_limits<float>::max();
        return score;
    }

    const fls_stats& stats() const { return stats_; }
    fls_handle handle() const { return handle_; }

private:
    static constexpr int kStride = static_cast<int>(sizeof(PCLPointXYZI) / sizeof(float));
    static const float* Data(const PCLPointCloudXYZI& c) { return c.points.empty() ? nullptr : reinterpret_cast<const float*>(c.points.data()); }
    static fls_params Blank() {
        fls_params p{};
        p.struct_size = sizeof(fls_params);
        return p;
    }
    fls_kind kind_;
    fls_handle handle_ = nullptr;
    fls_stats stats_{};
};

#endif  // FLS_HIP_REGISTRATION_H
