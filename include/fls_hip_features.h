// fls_hip_features.h -- header-only C++ drop-in for the two LOAM front-end classes of funny_lidar_slam on top of the
// C ABI of include/fls_features.h:
//
//     loam::PointcloudProjector::Project(PointcloudCluster&)        src/loam/pointcloud_projector.cpp:32-133
//     loam::FeatureExtractor::ExtractFeatures(PointcloudCluster&)   src/loam/feature_extractor.cpp:36-222
//
// One object replaces both (the projection stays on the device between the two calls).  Constructor arguments are
// the union of the two reference constructors (preprocessing.cpp:21-36).  De-skew is not done here: run
// LidarDistortionCorrector over raw_cloud_ first, or hand over the raw points when it is disabled.
// Needs only <lidar/pointcloud_cluster.h> of the reference (for PointcloudCluster) and fls_features.h.
#pragma once
#include "fls_features.h"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace loam {

class HipFeatureFrontEnd {
public:
    HipFeatureFrontEnd(int lidar_horizontal_scan, int lidar_vertical_scan, float lidar_horizontal_resolution, float min_distance,
                       float max_distance, float corner_thr, float planar_thr, int device = 0) {
        fls_feature_params p{};
        p.struct_size = sizeof(p);
        p.lidar_vertical_scan = lidar_vertical_scan;
        p.lidar_horizontal_scan = lidar_horizontal_scan;
        p.lidar_horizontal_resolution = lidar_horizontal_resolution;
        p.min_distance = min_distance;
        p.max_distance = max_distance;
        p.corner_thres = corner_thr;
        p.planar_thres = planar_thr;
        const fls_status rc = fls_features_create(&p, device, &h_);
        if (rc != FLS_OK) {  // the reference CHECK-aborts on unset parameters (pointcloud_projector.cpp:23-29, feature_extractor.cpp:20-23)
            std::fprintf(stderr, "HipFeatureFrontEnd: fls_features_create failed: %s\n", fls_status_string(rc));
            std::abort();
        }
    }
    ~HipFeatureFrontEnd() { fls_features_destroy(h_); }
    HipFeatureFrontEnd(const HipFeatureFrontEnd&) = delete;
    HipFeatureFrontEnd& operator=(const HipFeatureFrontEnd&) = delete;

    // fills ordered_cloud_, point_depth_vec_, point_col_index_vec_, row_start_index_vec_, row_end_index_vec_
    template <class Cluster>
    bool Project(Cluster& c) {
        using RawPoint = typename std::remove_reference<decltype(c.raw_cloud_.points[0])>::type;
        static const fls_point_layout lay{static_cast<uint32_t>(sizeof(RawPoint)), static_cast<uint32_t>(offsetof(RawPoint, x)),
                                          static_cast<uint32_t>(offsetof(RawPoint, intensity)), static_cast<uint32_t>(offsetof(RawPoint, ring))};
        size_t n = 0;
        const fls_status rc = fls_features_project(h_, c.raw_cloud_.points.data(), c.raw_cloud_.points.size(), &lay, &n);
        if (rc != FLS_OK) { std::fprintf(stderr, "HipFeatureFrontEnd::Project: %s\n", fls_status_string(rc)); return false; }
        FetchCloud(FLS_FEAT_ORDERED, c.ordered_cloud_);
        // the reference keeps the two vectors at rows * cols entries (pointcloud_projector.cpp:37-44); the first N are live
        if (c.point_depth_vec_.size() < n) c.point_depth_vec_.resize(n);
        if (c.point_col_index_vec_.size() < n) c.point_col_index_vec_.resize(n);
        fls_features_get(h_, FLS_FEAT_DEPTH, c.point_depth_vec_.data(), n);
        fls_features_get(h_, FLS_FEAT_COL, c.point_col_index_vec_.data(), n);
        c.row_start_index_vec_.resize(fls_features_get(h_, FLS_FEAT_ROW_START, nullptr, 0));
        c.row_end_index_vec_.resize(c.row_start_index_vec_.size());
        fls_features_get(h_, FLS_FEAT_ROW_START, c.row_start_index_vec_.data(), c.row_start_index_vec_.size());
        fls_features_get(h_, FLS_FEAT_ROW_END, c.row_end_index_vec_.data(), c.row_end_index_vec_.size());
        return true;
    }

    // fills corner_cloud_ and planar_cloud_ (before the two VoxelGrid filters of preprocessing.cpp:234-237)
    template <class Cluster>
    bool ExtractFeatures(Cluster& c) {
        size_t nc = 0, np = 0;
        const fls_status rc = fls_features_extract(h_, &nc, &np);
        if (rc != FLS_OK) { std::fprintf(stderr, "HipFeatureFrontEnd::ExtractFeatures: %s\n", fls_status_string(rc)); return false; }
        FetchCloud(FLS_FEAT_CORNER, c.corner_cloud_);
        FetchCloud(FLS_FEAT_PLANAR, c.planar_cloud_);
        return true;
    }

private:
    // library rows are packed {x, y, z, intensity}; pcl::PointXYZI is 32 bytes with the intensity in float 4
    template <class Cloud>
    void FetchCloud(int what, Cloud& out) {
        const size_t n = fls_features_get(h_, what, nullptr, 0);
        rows_.resize(4 * n);
        fls_features_get(h_, what, rows_.data(), n);
        out.points.resize(n);
        for (size_t k = 0; k < n; ++k) {
            auto& p = out.points[k];
            p.x = rows_[4 * k];
            p.y = rows_[4 * k + 1];
            p.z = rows_[4 * k + 2];
            p.intensity = rows_[4 * k + 3];
        }
    }
    fls_features_handle h_ = nullptr;
    std::vector<float> rows_;
};

}  // namespace loam
