/* ============================================================================
 * oracle/flo_api.h  --  TEST INFRASTRUCTURE ONLY (CPU oracle), C ABI for ctypes.
 *
 * The oracle is a CPU restatement of the reference's scan-to-map registration
 * path (include/registration/ headers, src/ivox_map/ sources).  It is NOT the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * liboracle.so.  The product (libfls_reg.so) never links or calls it.
 *
 * The struct layouts intentionally mirror include/fls_reg.h field-for-field so
 * the Python tests can drive both with one ctypes.Structure, but they are
 * declared independently here.
 * ==========================================================================*/
#ifndef FLO_API_H
#define FLO_API_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { FLO_ICP_OPTIMIZED = 0, FLO_P2PLANE_IVOX = 1, FLO_INCREMENTAL_NDT = 2, FLO_LOAM_FULL = 3,
       FLO_P2PLANE_KDTREE = 4 };

typedef struct flo_params {
    uint32_t struct_size;
    uint32_t max_iterations;
    int32_t is_localization_mode;
    uint32_t local_map_size;
    uint32_t local_corner_size;
    uint32_t local_planar_size;
    int32_t ndt_min_points_in_voxel;
    int32_t ndt_max_points_in_voxel;
    int32_t ndt_min_effective_pts;
    int32_t ndt_capacity;
    float map_cloud_filter_size;
    float source_cloud_filter_size;
    float corner_voxel_filter_size;
    float planar_voxel_filter_size;
    double point_to_planar_thres;
    double point_search_thres;
    double line_ratio_thres;
    double position_converge_thres;
    double rotation_converge_thres;
    double rot_thre_add_cloud;
    double dist_thre_add_cloud;
    double ndt_voxel_size;
    double ndt_res_outlier_threshold;
} flo_params;

typedef struct flo_stats {
    int32_t iterations;
    int32_t converged;
    int32_t n_valid;
    int32_t n_valid_corner;
    int32_t n_source;
    int32_t n_source_corner;
    int32_t map_updated;
    int32_t reserved;
    double sum_res;
    double sum_res_corner;
    double last_dx[6];
} flo_stats;

/* algorithmic-traffic counters of the last Match (SURVEY.md 8d formula inputs) */
typedef struct flo_counters {
    uint64_t point_iters;   /* source points x iterations actually executed        */
    uint64_t probes;        /* voxel / cell probes issued                          */
    uint64_t hit_voxels;    /* probes that found an occupied voxel / cell          */
    uint64_t cand_points;   /* map points scanned inside hit voxels / cells        */
    uint64_t tie_queries;   /* queries whose K-th/(K+1)-th candidate d2 tie exactly */
} flo_counters;

void* flo_create(int kind, const flo_params* p);
void flo_destroy(void* h);
void flo_set_threads(int n); /* OpenMP threads of the per-point stage (reduction stays sequential) */
int flo_get_threads(void);

/* AddCloudToLocalMap({cloud0[, cloud1]}) -- cloud0 = planar / ordered, cloud1 = corner */
int flo_add_cloud(void* h, const float* c0, size_t n0, const float* c1, size_t n1, int stride_floats);
/* Match(): src0 = ordered_cloud_ (ICP, NDT) or planar_cloud_ (others); src1 = corner_cloud_ (LoamFull).
 * update_map = 0 suppresses the map update inside Match (bench/parity of the pure registration). */
int flo_match(void* h, const float* src0, size_t n0, const float* src1, size_t n1, int stride_floats,
              double T_colmajor[16], int update_map, flo_stats* stats);
float flo_fitness(void* h, float max_range);

/* introspection for parity tests */
int flo_get_iteration_log(void* h, double* T_iters /*cap x 16*/, int32_t* n_valid, double* sum_res, int cap);
/* neighbour ids (map insertion ids) and counts held for each source point after the last Match.
 * slot: 0 planar/ordered, 1 corner.  ids: n x K int32 (K=5, or 1 for ICP, 7 voxel ids for NDT). */
int flo_get_correspondences(void* h, int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap_points);
int flo_get_counters(void* h, flo_counters* out);
/* per query of the last Match: 1 = its current neighbour list came out of a search with an exact distance tie (order / membership decided by
 * libstdc++'s introselect); returns the number of queries (0: this kind keeps no tie flags).  The GPU parity tests require every row that
 * differs from the oracle's to carry this flag. */
size_t flo_get_tie_flags(void* h, uint8_t* out, size_t cap);
/* 0: the kNN stage skips its traffic / tie bookkeeping (an extra vector + sort per query): what cpu_baseline times */
void flo_set_instrumentation(void* h, int on);
/* TEST SWITCH, process-wide, off by default: exact distance ties of the iVox kNN candidates ordered by insertion id (what the device's (d2, id) keys do)
 * instead of by libstdc++'s introselect permutation (what the reference does).  Only used to prove that a difference between the device and the
 * oracle is a tie and nothing else. */
void flo_set_tie_break_by_id(int on);
/* last Match's per-iteration H (36, col-major) and g (6) for reduction-tolerance tests */
int flo_get_last_system(void* h, double* H36, double* g6);
size_t flo_map_size(void* h, int slot); /* points (iVox / kd maps) or voxels (NDT) */
size_t flo_map_voxels(void* h);
void flo_set_ivox_capacity(void* h, size_t cap); /* test hook for the LRU eviction rule */
/* dump of the map in insertion-id order: xyz (n x 3) */
size_t flo_map_dump(void* h, int slot, float* xyz, size_t cap_points);
/* NDT voxel dump: keys (n x 3 int32), mu (n x 3), info (n x 9 col-major), estimated (n) */
size_t flo_ndt_dump(void* h, int32_t* keys, double* mu, double* info, uint8_t* est, int32_t* npts, size_t cap);

/* ---- LOAM feature front-end (src/loam/pointcloud_projector.cpp, src/loam/feature_extractor.cpp): see flo_features.h ---- */
typedef struct flo_feat_params {
    uint32_t struct_size;
    int32_t vertical_scan, horizontal_scan;
    float horizontal_resolution, min_distance, max_distance, corner_thres, planar_thres;
} flo_feat_params;
enum { FLO_FEAT_ORDERED = 0, FLO_FEAT_DEPTH = 1, FLO_FEAT_COL = 2, FLO_FEAT_ROW_START = 3, FLO_FEAT_ROW_END = 4, FLO_FEAT_CORNER = 5,
       FLO_FEAT_PLANAR = 6, FLO_FEAT_IS_CORNER = 7, FLO_FEAT_ROUGHNESS = 8, FLO_FEAT_VALID_PRE = 9, FLO_FEAT_VALID_POST = 10,
       FLO_FEAT_CORNER_IDX = 11, FLO_FEAT_PLANAR_IDX = 12, FLO_FEAT_RAW_INDEX = 13 };
void* flo_feat_create(const flo_feat_params* p);
void flo_feat_destroy(void* h);
/* Project(): raw points as a byte-strided AoS (x,y,z floats at off_xyz, intensity float at off_intensity, ring u16 at off_ring) */
int64_t flo_feat_project(void* h, const void* raw, size_t n, size_t stride_bytes, size_t off_xyz, size_t off_intensity, size_t off_ring);
int flo_feat_extract(void* h); /* 1 = ran, 0 = fewer than 12 ordered points */
/* copy one result array (element = 16 B xyzi, float, int32 or uint8 depending on `what`); returns its element count */
size_t flo_feat_get(void* h, int what, void* out, size_t cap_elems);
uint64_t flo_feat_tie_pairs(void* h);
/* 0 (default): equal-roughness points keep index order; 1: plain std::sort = the order libstdc++ gives the reference */
void flo_feat_set_sort_mode(void* h, int mode);
int flo_col_index(float x, float y, float h_res, int cols);
float flo_fast_atan2f(float y, float x);

/* ---- loop-closure matcher (src/slam/loop_closure.cpp:233-267: 4-resolution pcl NDT + pcl GICP + getFitnessScore): flo_loop.h ---- */
typedef struct flo_loop_stats {
    int32_t ndt_iterations[4], ndt_evaluations[4], ndt_source_points[4], ndt_target_leaves[4];
    int32_t gicp_iterations, gicp_inner_iterations, gicp_evaluations, gicp_correspondences, gicp_source_points, gicp_target_points, gicp_failed, reserved;
    double ndt_score[4];
    double T_after_ndt[16];
} flo_loop_stats;
float flo_loop_match(const float* src, size_t ns, const float* tgt, size_t nt, int stride_floats, double T_colmajor[16], flo_loop_stats* st);
/* pieces: NDT score / gradient / Hessian at pose vector p (x, y, z, rx, ry, rz) for the clouds AS GIVEN (leaf Gaussians at `resolution`) */
int flo_ndt_derivatives(const float* src, size_t ns, const float* tgt, size_t nt, int stride_floats, float resolution, const double p[6], double* score,
                        double grad[6], double hess[36]);
size_t flo_ndt_leaves(const float* tgt, size_t nt, int stride_floats, float resolution, int32_t* idx, int32_t* nr, double* mean3, double* icov9,
                      float* centroid3, size_t cap);
void flo_gicp_covariances(const float* c, size_t n, int stride_floats, int k, double eps, double* out9);
/* correspondences + Mahalanobis matrices for `guess` (transformation_ = identity), then f and g of the BFGS functor at x */
int flo_gicp_fdf(const float* src, size_t ns, const float* tgt, size_t nt, int stride_floats, const double guess[16], double corr_dist, const double x[6],
                 double* f, double g[6], int32_t* n_corr);
void flo_jacobi_svd_solve6(const double A[36], const double b[6], double x[6]);

/* stand-alone pieces for unit tests */
size_t flo_voxel_grid(const float* in, size_t n, int stride_floats, float leaf, float* out_xyzi /* n x 4 */);
void flo_so3_exp(const double v[3], double R_colmajor[9]);
void flo_so3_hat(const double v[3], double M_colmajor[9]);
void flo_rpy(const double R_colmajor[9], double rpy[3]);
void flo_colpiv_qr_solve_5x3(const double A_colmajor[15], const double b[5], double x[3]);
void flo_fullpiv_qr_solve_6(const double A_colmajor[36], const double b[6], double x[6]);
void flo_lu_inverse_6(const double A_colmajor[36], double inv[36], double* det);
void flo_inverse3(const double A[9], double inv[9]);
void flo_svd3(const double A[9], double U[9], double S[3], double V[9]);
int flo_knn_bruteforce(const float* map_xyz, size_t m, const float* q, int k, int32_t* idx, float* d2);
int flo_kdtree_knn(const float* map_xyz, size_t m, const float* queries, size_t nq, int k, int32_t* idx, float* d2);

#ifdef __cplusplus
}
#endif
#endif
