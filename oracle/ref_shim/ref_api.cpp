// ============================================================================
// oracle/ref_shim/ref_api.cpp  --  TEST INFRASTRUCTURE ONLY.
//
// C entry points over the REFERENCE'S OWN registration classes, compiled verbatim from
// /root/reference (nothing is copied into this repository; oracle/ref_shim/Makefile compiles the
// sources where they lie and writes oracle/_ref/libref.so):
//     include/registration/loam_point_to_plane_ivox.h   LoamPointToPlaneIVOX<double>
//     include/registration/icp_optimized.h              IcpOptimized<double>
//     include/registration/incremental_ndt.h            IncrementalNDT
//     include/registration/loam_full_kdtree.h           LoamFull<double>
//     include/registration/loam_point_to_plane_kdtree.h LoamPointToPlaneKdtree<double>
//     src/ivox_map/ivox_map.cpp, src/ivox_map/voxel_grid_node.cpp
//     src/loam/pointcloud_projector.cpp, src/loam/feature_extractor.cpp, src/lidar/lidar_model.cpp
//     include/common/{math_function,pointcloud_utility,hash_function,compare_function,...}.h
// against the include-shadow shim in oracle/ref_shim/include (Eigen / PCL / glog stand-ins; see eigen_shim.hpp for
// what that does and does not pin).  Constructed exactly as FrontEnd::InitMatcher does (src/slam/frontend.cpp:30-88).
//
// The reference keeps function-static state (is_first, last_T): one matcher of a kind per PROCESS --
// tests/refpin.py runs every scenario in a fresh worker process.
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

#include <Eigen/Dense>
#include <glog/logging.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/common/transforms.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/kdtree/kdtree_flann.h>

// introspection of the reference's private per-point state (flags, residuals, nearest_points_, maps): test harness only
#define private public
#define protected public
#include "registration/loam_point_to_plane_ivox.h"
#include "registration/icp_optimized.h"
#include "registration/incremental_ndt.h"
#include "registration/loam_full_kdtree.h"
#include "registration/loam_point_to_plane_kdtree.h"
#include "loam/pointcloud_projector.h"
#include "loam/feature_extractor.h"
#undef private
#undef protected

#include "../flo_api.h"

namespace {

using Cloud = PCLPointCloudXYZI;

Cloud make_cloud(const float* p, size_t n, int stride) {
    Cloud c;
    c.points.resize(n);
    for (size_t k = 0; k < n; ++k) {
        PCLPointXYZI q;
        q.x = p[k * stride]; q.y = p[k * stride + 1]; q.z = p[k * stride + 2];
        q.intensity = stride >= 8 ? p[k * stride + 4] : stride >= 4 ? p[k * stride + 3] : 0.0f;
        c.points[k] = q;
    }
    c.width = std::uint32_t(n); c.height = 1;
    return c;
}

struct Handle {
    int kind = -1;
    flo_params p{};
    std::shared_ptr<RegistrationInterface> m;
    LoamPointToPlaneIVOX<double>* ivox = nullptr;
    IcpOptimized<double>* icp = nullptr;
    IncrementalNDT* ndt = nullptr;
    LoamFull<double>* loam = nullptr;
    LoamPointToPlaneKdtree<double>* kd = nullptr;
    std::string error;
    int last_iters = -1;
};

// "num iter= <i>" is streamed by every matcher when its stop rule fires (loam_point_to_plane_ivox.h:184,
// icp_optimized.h:140, incremental_ndt.h:317, loam_full_kdtree.h:166, loam_point_to_plane_kdtree.h:132)
int iterations_from_log(int max_iter) {
    int it = -1;
    for (const std::string& l : ref_shim::log_lines()) {
        const size_t pos = l.find("num iter=");
        if (pos != std::string::npos) it = std::atoi(l.c_str() + pos + 9);
    }
    return it >= 0 ? it + 1 : max_iter;
}

template <class F>
int guarded(Handle* h, F&& f) {
    try { return f(); }
    catch (const std::exception& e) { if (h) h->error = e.what(); return -1; }
}

}  // namespace

extern "C" {

const char* ref_last_error(void* hh) { return static_cast<Handle*>(hh)->error.c_str(); }

// constructor argument order = the reference's ctors, values from flo_params (src/slam/frontend.cpp:30-88)
void* ref_create(int kind, const flo_params* p) {
    if (!p || p->struct_size != sizeof(flo_params)) return nullptr;
    auto* h = new Handle();
    h->kind = kind;
    h->p = *p;
    const bool loc = p->is_localization_mode != 0;
    const int rc = guarded(h, [&]() {
        switch (kind) {
            case FLO_P2PLANE_IVOX: {
                auto q = std::make_shared<LoamPointToPlaneIVOX<double>>(p->point_to_planar_thres, p->position_converge_thres,
                                                                        p->rotation_converge_thres, size_t(p->max_iterations), loc);
                h->ivox = q.get(); h->m = q; break;
            }
            case FLO_ICP_OPTIMIZED: {
                auto q = std::make_shared<IcpOptimized<double>>(p->max_iterations, p->local_map_size, p->map_cloud_filter_size,
                                                                p->source_cloud_filter_size, p->point_search_thres, p->position_converge_thres,
                                                                p->rotation_converge_thres, p->rot_thre_add_cloud, p->dist_thre_add_cloud, loc);
                h->icp = q.get(); h->m = q; break;
            }
            case FLO_INCREMENTAL_NDT: {
                auto q = std::make_shared<IncrementalNDT>(p->ndt_voxel_size, p->ndt_res_outlier_threshold, p->source_cloud_filter_size,
                                                          p->rotation_converge_thres, p->position_converge_thres, p->ndt_min_points_in_voxel,
                                                          p->ndt_max_points_in_voxel, p->ndt_min_effective_pts, p->ndt_capacity,
                                                          int(p->max_iterations), loc);
                h->ndt = q.get(); h->m = q; break;
            }
            case FLO_LOAM_FULL: {
                auto q = std::make_shared<LoamFull<double>>(p->point_to_planar_thres, p->point_search_thres, p->line_ratio_thres,
                                                            p->position_converge_thres, p->rotation_converge_thres, p->dist_thre_add_cloud,
                                                            p->rot_thre_add_cloud, size_t(p->local_corner_size), size_t(p->local_planar_size),
                                                            p->corner_voxel_filter_size, p->planar_voxel_filter_size, int(p->max_iterations));
                h->loam = q.get(); h->m = q; break;
            }
            case FLO_P2PLANE_KDTREE: {
                auto q = std::make_shared<LoamPointToPlaneKdtree<double>>(p->point_to_planar_thres, p->position_converge_thres,
                                                                          p->rotation_converge_thres, p->rot_thre_add_cloud, p->dist_thre_add_cloud,
                                                                          p->local_map_size, p->map_cloud_filter_size, size_t(p->max_iterations), loc);
                h->kd = q.get(); h->m = q; break;
            }
            default: return -1;
        }
        return 0;
    });
    if (rc != 0) { std::fprintf(stderr, "[ref] create failed: %s\n", h->error.c_str()); delete h; return nullptr; }
    return h;
}

void ref_destroy(void* hh) { delete static_cast<Handle*>(hh); }

// test hook for the iVox LRU rule (the reference hard-codes capacity_ = 1e6, ivox_map.h:36)
void ref_set_ivox_capacity(void* hh, size_t cap) {
    auto* h = static_cast<Handle*>(hh);
    if (h->ivox) h->ivox->ivox_map_ptr_->options_.capacity_ = cap;
}

int ref_add_cloud(void* hh, const float* c0, size_t n0, const float* c1, size_t n1, int stride) {
    auto* h = static_cast<Handle*>(hh);
    return guarded(h, [&]() {
        if (h->kind == FLO_LOAM_FULL) h->m->AddCloudToLocalMap({make_cloud(c0, n0, stride), make_cloud(c1, n1, stride)});
        else h->m->AddCloudToLocalMap({make_cloud(c0, n0, stride)});
        return 0;
    });
}

// returns 0 = Match() true, 1 = false, -1 = CHECK failure / exception
int ref_match(void* hh, const float* s0, size_t n0, const float* s1, size_t n1, int stride, double T_colmajor[16], flo_stats* st) {
    auto* h = static_cast<Handle*>(hh);
    return guarded(h, [&]() {
        auto cluster = std::make_shared<PointcloudCluster>();
        if (h->kind == FLO_ICP_OPTIMIZED || h->kind == FLO_INCREMENTAL_NDT) cluster->ordered_cloud_ = make_cloud(s0, n0, stride);
        else cluster->planar_cloud_ = make_cloud(s0, n0, stride);
        if (h->kind == FLO_LOAM_FULL) cluster->corner_cloud_ = make_cloud(s1, n1, stride);
        Mat4d T;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) T(i, j) = T_colmajor[i + 4 * j];
        ref_shim::log_lines().clear();
        const bool ok = h->m->Match(cluster, T);
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) T_colmajor[i + 4 * j] = T(i, j);
        h->last_iters = iterations_from_log(int(h->p.max_iterations));
        if (st) {
            std::memset(st, 0, sizeof(*st));
            st->iterations = h->last_iters;
            st->converged = ok ? 1 : 0;
            if (h->ivox) { st->n_valid = int(h->ivox->number_valid_planar_); st->sum_res = h->ivox->overall_res_planar_; st->n_source = int(h->ivox->number_planar_point_); }
            if (h->kd) { st->n_valid = int(h->kd->number_valid_planar_); st->sum_res = h->kd->overall_res_planar_; st->n_source = int(h->kd->number_planar_point_); }
            if (h->loam) {
                st->n_valid = int(h->loam->number_valid_planar_); st->n_valid_corner = int(h->loam->number_valid_corner_);
                st->sum_res = h->loam->overall_res_planar_; st->sum_res_corner = h->loam->overall_res_corner_;
                st->n_source = int(h->loam->number_planar_point_); st->n_source_corner = int(h->loam->number_corner_point_);
            }
            if (h->icp) st->n_source = int(h->icp->source_cloud_ptr_->size());
            if (h->ndt) st->n_source = int(h->ndt->source_cloud->size());
        }
        return ok ? 0 : 1;
    });
}

float ref_fitness(void* hh, float max_range) {
    auto* h = static_cast<Handle*>(hh);
    float f = -1.0f;
    guarded(h, [&]() { f = h->m->GetFitnessScore(max_range); return 0; });
    return f;
}

// ---- introspection ---------------------------------------------------------------------------------------------
static const Cloud* map_cloud(Handle* h, int slot) {
    if (h->icp) return h->icp->local_map_ptr_.get();
    if (h->kd) return h->kd->local_map_ptr_.get();
    if (h->loam) return slot == 1 ? h->loam->local_corner_cloud_ptr_.get() : h->loam->local_planar_cloud_ptr_.get();
    return nullptr;
}

// points of the local map: kd-tree kinds in cloud order (= the kd-tree's index space); iVox in LRU-list order (front =
// most recently touched voxel), points of a voxel in insertion order; NDT: number of voxels
size_t ref_map_size(void* hh, int slot) {
    auto* h = static_cast<Handle*>(hh);
    if (const Cloud* c = map_cloud(h, slot)) return c->size();
    if (h->ivox) { size_t n = 0; for (auto& kv : h->ivox->ivox_map_ptr_->grids_cache_) n += kv.second.Size(); return n; }
    if (h->ndt) return h->ndt->data_.size();
    return 0;
}
size_t ref_map_voxels(void* hh) {
    auto* h = static_cast<Handle*>(hh);
    if (h->ivox) return h->ivox->ivox_map_ptr_->grids_cache_.size();
    if (h->ndt) return h->ndt->data_.size();
    return 0;
}
// xyzi rows; for iVox additionally the voxel key of every point (keys may be NULL)
size_t ref_map_dump(void* hh, int slot, float* xyzi, int32_t* keys, size_t cap) {
    auto* h = static_cast<Handle*>(hh);
    size_t k = 0;
    if (const Cloud* c = map_cloud(h, slot)) {
        for (; k < std::min(cap, c->size()); ++k) { xyzi[4 * k] = c->points[k].x; xyzi[4 * k + 1] = c->points[k].y; xyzi[4 * k + 2] = c->points[k].z; xyzi[4 * k + 3] = c->points[k].intensity; }
        return c->size();
    }
    if (h->ivox) {
        for (auto& kv : h->ivox->ivox_map_ptr_->grids_cache_)
            for (size_t i = 0; i < kv.second.Size(); ++i, ++k) {
                if (k >= cap) continue;
                const PCLPointXYZI q = kv.second.GetPoint(i);
                xyzi[4 * k] = q.x; xyzi[4 * k + 1] = q.y; xyzi[4 * k + 2] = q.z; xyzi[4 * k + 3] = q.intensity;
                if (keys) { keys[3 * k] = kv.first[0]; keys[3 * k + 1] = kv.first[1]; keys[3 * k + 2] = kv.first[2]; }
            }
        return k;
    }
    return 0;
}

// per-point state after the last Match: valid flag and |residual| (slot 0 planar, 1 corner); ICP / NDT keep theirs in locals
int ref_get_flags(void* hh, int slot, uint8_t* valid, double* res, size_t cap) {
    auto* h = static_cast<Handle*>(hh);
    const std::vector<bool>* f = nullptr; const std::vector<double>* r = nullptr;
    if (h->ivox) { f = &h->ivox->planar_valid_flags_; r = &h->ivox->res_planars_; }
    else if (h->kd) { f = &h->kd->planar_valid_flags_; r = &h->kd->res_planars_; }
    else if (h->loam) { f = slot == 1 ? &h->loam->corner_valid_flags_ : &h->loam->planar_valid_flags_; r = slot == 1 ? &h->loam->res_corners_ : &h->loam->res_planars_; }
    if (!f) return -1;
    const size_t n = std::min(cap, f->size());
    for (size_t i = 0; i < n; ++i) { valid[i] = (*f)[i] ? 1 : 0; res[i] = (*f)[i] ? (*r)[i] : 0.0; }
    return int(f->size());
}
// iVox nearest_points_ (5 x xyz per source point, count per point), in the reference's own slot order
int ref_get_nearest(void* hh, float* xyz, uint8_t* cnt, size_t cap) {
    auto* h = static_cast<Handle*>(hh);
    if (!h->ivox) return -1;
    const auto& np = h->ivox->nearest_points_;
    const size_t n = std::min(cap, np.size());
    for (size_t i = 0; i < n; ++i) {
        cnt[i] = uint8_t(std::min<size_t>(np[i].size(), 255));
        for (size_t j = 0; j < 5; ++j)
            for (int a = 0; a < 3; ++a) xyz[(i * 5 + j) * 3 + a] = j < np[i].size() ? np[i][j].data[a] : 0.0f;
    }
    return int(np.size());
}
int ref_last_system(void* hh, double* H36, double* g6) {
    auto* h = static_cast<Handle*>(hh);
    const Eigen::Matrix<double, 6, 6>* H = nullptr; const Eigen::Matrix<double, 6, 1>* g = nullptr;
    if (h->ivox) { H = &h->ivox->H_; g = &h->ivox->g_; }
    else if (h->kd) { H = &h->kd->H_; g = &h->kd->g_; }
    else if (h->loam) { H = &h->loam->H_; g = &h->loam->g_; }
    if (!H) return -1;
    std::memcpy(H36, H->data(), 36 * sizeof(double));
    std::memcpy(g6, g->data(), 6 * sizeof(double));
    return 0;
}
// NDT voxels in LRU-list order (front first): key, mu, sigma, information (col-major), estimated, num_points_, pending points
size_t ref_ndt_dump(void* hh, int32_t* keys, double* mu, double* sigma, double* info, uint8_t* est, int32_t* npts, int32_t* pending, size_t cap) {
    auto* h = static_cast<Handle*>(hh);
    if (!h->ndt) return 0;
    size_t k = 0;
    for (const auto& kd : h->ndt->data_) {
        if (k < cap) {
            for (int a = 0; a < 3; ++a) { keys[3 * k + a] = kd.first[a]; mu[3 * k + a] = kd.second.mu_[a]; }
            std::memcpy(sigma + 9 * k, kd.second.sigma_.data(), 72);
            std::memcpy(info + 9 * k, kd.second.information_.data(), 72);
            est[k] = kd.second.ndt_estimated_ ? 1 : 0;
            npts[k] = kd.second.num_points_;
            pending[k] = int32_t(kd.second.points_.size());
        }
        ++k;
    }
    return k;
}

// ---- LOAM feature front-end: PointcloudProjector::Project + FeatureExtractor::ExtractFeatures, compiled verbatim --------
struct RefFeat {
    std::unique_ptr<loam::PointcloudProjector> proj;
    std::unique_ptr<loam::FeatureExtractor> feat;
    PointcloudCluster cluster;
    int rows = 0, cols = 0;
};
void* ref_feat_create(const flo_feat_params* p) {
    if (!p || p->struct_size != sizeof(flo_feat_params)) return nullptr;
    auto* s = new RefFeat();
    try {
        s->rows = p->vertical_scan; s->cols = p->horizontal_scan;
        s->proj.reset(new loam::PointcloudProjector(std::make_shared<LidarDistortionCorrector>(), p->horizontal_scan, p->vertical_scan,
                                                    p->horizontal_resolution, p->min_distance, p->max_distance));
        s->feat.reset(new loam::FeatureExtractor(p->corner_thres, p->planar_thres, p->horizontal_scan, p->vertical_scan));
    } catch (const std::exception& e) { std::fprintf(stderr, "[ref] feat create: %s\n", e.what()); delete s; return nullptr; }
    return s;
}
void ref_feat_destroy(void* h) { delete static_cast<RefFeat*>(h); }
int64_t ref_feat_project(void* hh, const void* raw, size_t n, size_t stride_bytes, size_t off_xyz, size_t off_intensity, size_t off_ring, size_t off_time) {
    auto* s = static_cast<RefFeat*>(hh);
    try {
        s->cluster.raw_cloud_.points.resize(n);
        const char* b = static_cast<const char*>(raw);
        for (size_t i = 0; i < n; ++i) {
            PointXYZIRT q{};
            float xyz[3]; std::memcpy(xyz, b + i * stride_bytes + off_xyz, 12);
            q.x = xyz[0]; q.y = xyz[1]; q.z = xyz[2];
            std::memcpy(&q.intensity, b + i * stride_bytes + off_intensity, 4);
            std::uint16_t ring; std::memcpy(&ring, b + i * stride_bytes + off_ring, 2);
            q.ring = std::uint8_t(ring);
            std::memcpy(&q.time, b + i * stride_bytes + off_time, 4);
            s->cluster.raw_cloud_.points[i] = q;
        }
        s->proj->Project(s->cluster);
        return int64_t(s->cluster.ordered_cloud_.size());
    } catch (const std::exception& e) { std::fprintf(stderr, "[ref] project: %s\n", e.what()); return -1; }
}
int ref_feat_extract(void* hh) {
    auto* s = static_cast<RefFeat*>(hh);
    try { s->feat->ExtractFeatures(s->cluster); return 1; }
    catch (const std::exception& e) { std::fprintf(stderr, "[ref] extract: %s\n", e.what()); return -1; }
}
// same `what` codes as flo_feat_get (flo_api.h); arrays the reference does not keep return 0
size_t ref_feat_get(void* hh, int what, void* out, size_t cap) {
    auto* s = static_cast<RefFeat*>(hh);
    auto cloud = [&](const Cloud& c) {
        float* o = static_cast<float*>(out);
        for (size_t i = 0; i < std::min(cap, c.size()); ++i) { o[4 * i] = c.points[i].x; o[4 * i + 1] = c.points[i].y; o[4 * i + 2] = c.points[i].z; o[4 * i + 3] = c.points[i].intensity; }
        return c.size();
    };
    const size_t n = s->cluster.ordered_cloud_.size();
    switch (what) {
        case FLO_FEAT_ORDERED: return cloud(s->cluster.ordered_cloud_);
        case FLO_FEAT_CORNER: return cloud(s->cluster.corner_cloud_);
        case FLO_FEAT_PLANAR: return cloud(s->cluster.planar_cloud_);
        case FLO_FEAT_DEPTH: std::memcpy(out, s->cluster.point_depth_vec_.data(), std::min(cap, n) * 4); return n;
        case FLO_FEAT_COL: std::memcpy(out, s->cluster.point_col_index_vec_.data(), std::min(cap, n) * 4); return n;
        case FLO_FEAT_ROW_START: std::memcpy(out, s->cluster.row_start_index_vec_.data(), std::min(cap, size_t(s->rows)) * 4); return size_t(s->rows);
        case FLO_FEAT_ROW_END: std::memcpy(out, s->cluster.row_end_index_vec_.data(), std::min(cap, size_t(s->rows)) * 4); return size_t(s->rows);
        case FLO_FEAT_IS_CORNER: for (size_t i = 0; i < std::min(cap, n); ++i) static_cast<uint8_t*>(out)[i] = s->feat->is_corners_[i] ? 1 : 0; return n;
        case FLO_FEAT_VALID_POST: for (size_t i = 0; i < std::min(cap, n); ++i) static_cast<uint8_t*>(out)[i] = s->feat->is_valid_points_[i] ? 1 : 0; return n;
        default: return 0;
    }
}
int ref_col_index(float x, float y, const char* lidar_type) { return LidarModel::Instance(lidar_type)->ColIndex(x, y); }
float ref_fast_atan2f(float y, float x) { return FastAtan2(y, x); }
// threads of the parallel-STL loops: > 0 only in the libref_par.so build (include/pstl_omp.hpp); 0 = the serial PSTL backend
int ref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 0;
#endif
}
void ref_so3_exp(const double v[3], double R_colmajor[9]) {
    const Eigen::Matrix<double, 3, 1> w(v[0], v[1], v[2]);
    const Eigen::Matrix<double, 3, 3> R = SO3Exp(w);
    std::memcpy(R_colmajor, R.data(), 72);
}
void ref_rpy(const double R_colmajor[9], double rpy[3]) {
    Eigen::Matrix<double, 3, 3> R;
    std::memcpy(R.data(), R_colmajor, 72);
    const Eigen::Matrix<double, 3, 1> e = RotationMatrixToRPY(R);
    rpy[0] = e[0]; rpy[1] = e[1]; rpy[2] = e[2];
}

}  // extern "C"
