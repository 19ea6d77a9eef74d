/* ============================================================================
 * fls_features.h -- C ABI of the LOAM feature front-end on MI355X (gfx950): the step immediately before Match in
 * LoamFull_KdTree mode, same shared library as fls_reg.h (libfls_reg.so).
 *
 *   loam::PointcloudProjector::Project(PointcloudCluster&)        src/loam/pointcloud_projector.cpp:32-133
 *       -> fls_features_project    (range-image projection, first return per cell, ordered cloud, depth / column
 *                                   vectors, per-ring start / end indices)
 *   loam::FeatureExtractor::ExtractFeatures(PointcloudCluster&)   src/loam/feature_extractor.cpp:36-222
 *       -> fls_features_extract    (occlusion / parallel-beam marks, roughness, per-sector corner / planar selection)
 *   the two pcl::VoxelGrid filters preprocessing.cpp:234-237 applies to the feature clouds
 *       -> FLS_FEAT_CORNER_FILTERED / FLS_FEAT_PLANAR_FILTERED (host VoxelGrid, leaf sizes from the parameters)
 *
 * Constructor arguments = the reference's (preprocessing.cpp:21-36).  De-skew (LidarDistortionCorrector, IMU driven)
 * is not part of this library: hand over corrected points, or raw points when the sensor did not move.
 * Plain C; no exception crosses the boundary; a handle is not thread-safe.  No CPU fallback.
 * ==========================================================================*/
#ifndef FLS_FEATURES_H
#define FLS_FEATURES_H
#include "fls_reg.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fls_features* fls_features_handle;

typedef struct fls_feature_params {
    uint32_t struct_size;            /* = sizeof(fls_feature_params) */
    int32_t lidar_vertical_scan;     /* rings (rows),    LidarModel::vertical_scan_num_                      */
    int32_t lidar_horizontal_scan;   /* columns,         LidarModel::horizon_scan_num_  (<= 4096)            */
    float lidar_horizontal_resolution; /* rad,           LidarModel::h_res_                                  */
    float min_distance, max_distance;  /* lidar_use_min_dist_ / lidar_use_max_dist_                          */
    float corner_thres, planar_thres;  /* loam_feature_corner_thres_ / loam_feature_planar_thres_            */
    float corner_voxel_filter_size, planar_voxel_filter_size; /* 0 = no filtered cloud                       */
} fls_feature_params;

/* byte layout of one raw driver point; VelodynePointXYZIRT (lidar_point_type.h): {32, 0, 16, 20} */
typedef struct fls_point_layout {
    uint32_t stride_bytes, xyz_offset /* 3 floats */, intensity_offset /* float */, ring_offset /* uint16 */;
} fls_point_layout;

/* arrays fls_features_get returns (element: xyzi = 4 floats) */
enum {
    FLS_FEAT_ORDERED = 0,     /* xyzi   ordered_cloud_                                    */
    FLS_FEAT_DEPTH = 1,       /* float  point_depth_vec_[0 .. N)                          */
    FLS_FEAT_COL = 2,         /* int32  point_col_index_vec_[0 .. N)                      */
    FLS_FEAT_ROW_START = 3,   /* int32  row_start_index_vec_                              */
    FLS_FEAT_ROW_END = 4,     /* int32  row_end_index_vec_                                */
    FLS_FEAT_CORNER = 5,      /* xyzi   corner_cloud_ (before the voxel filter)           */
    FLS_FEAT_PLANAR = 6,      /* xyzi   planar_cloud_ (before the voxel filter)           */
    FLS_FEAT_IS_CORNER = 7,   /* uint8  is_corners_                                       */
    FLS_FEAT_ROUGHNESS = 8,   /* float  point_features_[i].roughness_ by ordered index    */
    FLS_FEAT_VALID_PRE = 9,   /* uint8  is_valid_points_ after SelectValidPoints          */
    FLS_FEAT_VALID_POST = 10, /* uint8  is_valid_points_ after SelectFeatures             */
    FLS_FEAT_CORNER_IDX = 11, /* int32  ordered-cloud index of every corner point         */
    FLS_FEAT_PLANAR_IDX = 12, /* int32  ordered-cloud index of every planar point         */
    FLS_FEAT_RAW_INDEX = 13,  /* int32  raw-cloud index of every ordered point            */
    FLS_FEAT_CORNER_FILTERED = 14, /* xyzi  VoxelGrid(corner_cloud_, corner_voxel_filter_size) */
    FLS_FEAT_PLANAR_FILTERED = 15  /* xyzi  VoxelGrid(planar_cloud_, planar_voxel_filter_size) */
};

/* PointcloudProjector + FeatureExtractor constructors (preprocessing.cpp:21-36); unset (FLT_MAX / INT_MAX) -> FLS_ERR_INVALID */
fls_status fls_features_create(const fls_feature_params* params, int device_id, fls_features_handle* out);
void fls_features_destroy(fls_features_handle h);
/* PointcloudProjector::Project: raw_cloud_.points.data() + layout; *n_ordered = ordered_cloud_.size() */
fls_status fls_features_project(fls_features_handle h, const void* raw_points, size_t n, const fls_point_layout* layout, size_t* n_ordered);
/* FeatureExtractor::ExtractFeatures on the last projection (device resident); FLS_ERR_STATE without one.
 * Fewer than 12 ordered points (the reference indexes N-6 .. N-1 and i-5 .. i+6): both clouds empty, FLS_OK.      */
fls_status fls_features_extract(fls_features_handle h, size_t* n_corner, size_t* n_planar);
/* copy a result array into `out` (NULL: only the size); returns its element count */
size_t fls_features_get(fls_features_handle h, int what, void* out, size_t cap_elems);
/* device time of the last project / extract call [ms] (hipEvents on the handle's stream) */
fls_status fls_features_get_time(fls_features_handle h, double* project_ms, double* extract_ms);

#ifdef __cplusplus
}
#endif
#endif
