/* ============================================================================
 * fls_reg.h -- C ABI of libfls_reg.so, the MI355X (gfx950) scan-to-map
 * registration back-end for funny_lidar_slam's RegistrationInterface.
 *
 * Drop-in boundary: the reference's plug-in interface is
 *
 *   class RegistrationInterface {                      // include/registration/registration_interface.h:11-20
 *     virtual bool  Match(const PointcloudClusterPtr&, Mat4d& T) = 0;                          // :13
 *     virtual void  AddCloudToLocalMap(const std::initializer_list<PCLPointCloudXYZI>&) = 0;   // :17
 *     virtual float GetFitnessScore(float max_range) const = 0;                                // :19
 *   };
 *
 * Each entry point below replaces one of those virtuals (or one constructor)
 * of the five implementations; the header-only adapter
 * include/fls_hip_registration.h turns them back into a RegistrationInterface.
 * Plain C: pointers + sizes only, no Eigen / PCL / torch types.  No exception
 * crosses this boundary; a handle is NOT thread-safe (same contract as the
 * reference: one owner thread, frontend.cpp:207-210 / localization.cpp:246-249).
 *
 * Clouds are passed as float AoS with a stride in floats: pcl::PointXYZI is
 * 32 B (x, y, z, pad | intensity, pad x3) => pass cloud.points.data() with
 * stride_floats = 8 (intensity is read from float 4); a packed xyz array uses 3,
 * xyzi uses 4 (intensity in float 3).  Poses are Eigen::Matrix4d::data(): 16 doubles,
 * COLUMN-major, world <- body.
 *
 * The library has no CPU fallback: every entry point that computes fails with
 * FLS_ERR_DEVICE when no gfx950 device is usable.
 * ==========================================================================*/
#ifndef FLS_REG_H
#define FLS_REG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLS_ABI_VERSION 1
/* Additive revision of ABI version 1: entry points are only ever ADDED under one FLS_ABI_VERSION (existing signatures, struct layouts and
 * status codes do not change), and this number counts the additions -- 1: fls_match_batch, map export / import; 2: fls_voxel_grid_cloud,
 * fls_features_*; 3: fls_replicas_*, fls_loop_match; 4: fls_debug_exact_sort; 5: fls_scan_upload_raw; 6: fls_map_image_*; 7: fls_debug_ldlt6.  A caller built against revision r works with any library
 * whose fls_abi_revision() >= r. */
#define FLS_ABI_REVISION 7

/* Which reference class the handle replaces (mode strings: include/common/constant_variable.h:21-25). */
typedef enum fls_kind {
    FLS_ICP_OPTIMIZED = 0,   /* IcpOptimized<double>            include/registration/icp_optimized.h:15          */
    FLS_P2PLANE_IVOX = 1,    /* LoamPointToPlaneIVOX<double>    include/registration/loam_point_to_plane_ivox.h:30 */
    FLS_INCREMENTAL_NDT = 2, /* IncrementalNDT                  include/registration/incremental_ndt.h:16        */
    FLS_LOAM_FULL = 3,       /* LoamFull<double>                include/registration/loam_full_kdtree.h:24       */
    FLS_P2PLANE_KDTREE = 4   /* LoamPointToPlaneKdtree<double>  include/registration/loam_point_to_plane_kdtree.h:25 */
} fls_kind;

typedef enum fls_status {
    FLS_OK = 0,             /* Match() == true                                                   */
    FLS_NOT_CONVERGED = 1,  /* Match() == false (T is still written, as in the reference)        */
    FLS_SKIPPED = 2,        /* fls_match_batch status[] only: job not run (an earlier job of its lane failed) */
    FLS_ERR_INVALID = -1,   /* bad argument / parameter left at its "unset" sentinel (CHECK_NE)  */
    FLS_ERR_DEVICE = -2,    /* HIP runtime error or no gfx950 device                             */
    FLS_ERR_RANGE = -3,     /* coordinate outside the +-2^20 voxel key range                     */
    FLS_ERR_NOMEM = -4,
    FLS_ERR_STATE = -5      /* call order violated (e.g. Match before any map cloud)             */
} fls_status;

/* Superset of the five constructors' argument lists (src/slam/frontend.cpp:30-88,
 * src/slam/localization.cpp:43-92).  Unused fields are ignored by a kind.
 *   P2PLANE_IVOX   : point_to_planar_thres, position/rotation_converge_thres, max_iterations, is_localization_mode
 *                    (iVox resolution 0.5 / NEARBY18 / capacity 1e6 / map filter 0.5 are hard-coded in the
 *                     reference, loam_point_to_plane_ivox.h:53-58,351, and therefore here)
 *   ICP_OPTIMIZED  : max_iterations, local_map_size, map_cloud_filter_size, source_cloud_filter_size,
 *                    point_search_thres (= max_correspond_distance, a SQUARED distance), position/rotation_converge_thres,
 *                    rot_thre_add_cloud, dist_thre_add_cloud, is_localization_mode
 *   INCREMENTAL_NDT: ndt_voxel_size, ndt_res_outlier_threshold, source_cloud_filter_size, rotation/position_converge_thres,
 *                    ndt_min/max_points_in_voxel, ndt_min_effective_pts, ndt_capacity, max_iterations, is_localization_mode
 *   LOAM_FULL      : point_to_planar_thres, point_search_thres, line_ratio_thres, position/rotation_converge_thres,
 *                    dist_thre_add_cloud, rot_thre_add_cloud, local_corner_size, local_planar_size,
 *                    corner/planar_voxel_filter_size, max_iterations
 *   P2PLANE_KDTREE : point_to_planar_thres, position/rotation_converge_thres, rot/dist_thre_add_cloud, local_map_size,
 *                    map_cloud_filter_size, max_iterations, is_localization_mode                                     */
typedef struct fls_params {
    uint32_t struct_size; /* = sizeof(fls_params); ABI guard */
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
} fls_params;

/* What the reference only prints in its DLOG line (icp_optimized.h:140-144 etc.). */
typedef struct fls_stats {
    int32_t iterations;      /* Gauss-Newton iterations executed                         */
    int32_t converged;       /* Match() return value                                     */
    int32_t n_valid;         /* valid planar / effective points of the last iteration    */
    int32_t n_valid_corner;  /* LOAM_FULL: valid corner points                           */
    int32_t n_source;        /* source points after the in-Match VoxelGrid (ICP, NDT)    */
    int32_t n_source_corner;
    int32_t map_updated;     /* 1 if Match() went on to AddCloudToLocalMap               */
    int32_t reserved;
    double sum_res;          /* overall_res_planar_ / total_res                          */
    double sum_res_corner;
    double last_dx[6];       /* last Gauss-Newton step, in the matcher's own state order */
} fls_stats;

typedef struct fls_matcher* fls_handle;

/* ---- constructors / destructor ---------------------------------------------------------------
 * replaces std::make_shared<Impl<double>>(...) at frontend.cpp:32,44,59,71 / localization.cpp:45,57,70,77 */
fls_status fls_create(fls_kind kind, const fls_params* params, int device_id, fls_handle* out);
void fls_destroy(fls_handle h);

/* ---- RegistrationInterface::AddCloudToLocalMap  (registration_interface.h:17) ------------------
 * cloud0 = the single cloud (ICP / NDT / P2PLANE_*: world frame on external calls) or the PLANAR cloud
 * (LOAM_FULL); cloud1 = the CORNER cloud (LOAM_FULL only, else NULL / 0) -- the initializer_list order
 * of loam_full_kdtree.h:67-69.                                                                          */
fls_status fls_add_cloud_to_local_map(fls_handle h, const float* cloud0, size_t n0, const float* cloud1, size_t n1,
                                      int stride_floats);

/* ---- RegistrationInterface::Match  (registration_interface.h:13) -------------------------------
 * src0 = cluster->ordered_cloud_ (ICP, NDT) or cluster->planar_cloud_ (others); src1 = cluster->corner_cloud_
 * (LOAM_FULL).  T_colmajor is in/out.  update_map != 0 reproduces the reference (Match itself grows the
 * local map in mapping mode); 0 skips that step (benchmarks time the registration alone).
 * Returns FLS_OK (true), FLS_NOT_CONVERGED (false) or an error (< 0).                                   */
fls_status fls_match(fls_handle h, const float* src0, size_t n0, const float* src1, size_t n1, int stride_floats,
                     double T_colmajor[16], int update_map, fls_stats* stats);

/* ---- batch of independent registrations against the handle's CURRENT map (BASELINE configs[4]; SURVEY.md 8b "batch",
 * 8e).  Job j is what a fresh reference matcher holding this map returns for Match(src0[j] (, src1[j]), T[16 j ..]):
 * no map update, no state carried between jobs (the reference's function-static / per-instance state is per job,
 * SURVEY Q12).  `lanes` (1..16) registrations are kept in flight on separate HIP streams (clones of the handle that
 * read its resident map; every kind; lanes <= 1 = one clone, jobs back to back).  T is n_jobs x 16 doubles, column-major,
 * in/out; stats / status (per-job fls_status, FLS_SKIPPED for a job that was not run because an earlier job of its lane
 * failed) may be NULL; src1 and n1 are both NULL unless the kind takes a second cloud.  Returns the first error (< 0) or
 * FLS_OK.  The handle's own Match state (nearest_points_, keyframe gate, resident scan, last pose) is never touched.    */
fls_status fls_match_batch(fls_handle h, size_t n_jobs, const float* const* src0, const size_t* n0, const float* const* src1,
                           const size_t* n1, int stride_floats, double* T_colmajor, fls_stats* stats, int32_t* status, int lanes);

/* ---- map image export / import (BASELINE configs[4], SURVEY.md 8e: rank 0 builds the map, the image is broadcast, every
 * GPU imports it instead of re-inserting 1e6 points).  The blob is self-contained host memory (voxel keys in LRU order, the
 * points with their insertion ids, counters): import into a handle of the same kind created with the same parameters gives a
 * handle whose Match / AddCloudToLocalMap results are identical to the exporter's from then on.  FLS_P2PLANE_IVOX only
 * (other kinds: export returns 0, import FLS_ERR_STATE).  fls_map_export(h, NULL, 0) returns the size needed.            */
size_t fls_map_export(fls_handle h, void* blob, size_t cap_bytes);
fls_status fls_map_import(fls_handle h, const void* blob, size_t n_bytes);
/* The DEVICE image itself as one flat buffer (revision 6): what a torch.distributed rank broadcasts to the others in place of the blob above
 * ("rank 0 builds the device map image, one broadcast over xGMI").  `dst` / `src` is memory of the handle's own device when
 * *_on_device != 0 -- e.g. the data_ptr of the CUDA tensor an RCCL broadcast works on -- or host memory (pinned for speed; gloo).  The exporter
 * keeps its map.  The importer becomes a READ-ONLY REPLICA, like a member of fls_replicas_*: fls_match / fls_match_batch with update_map == 0;
 * fls_map_import or a replica refresh makes it an ordinary handle again.  Header and contents are validated on import (sizes against n_bytes,
 * every cell inside the point array): FLS_ERR_INVALID otherwise.  Only the iVox kind has such an image (others: 0 / FLS_ERR_STATE). */
size_t fls_map_image_bytes(fls_handle h);
fls_status fls_map_image_export(fls_handle h, void* dst, size_t cap_bytes, int dst_on_device);
fls_status fls_map_image_import(fls_handle h, const void* src, size_t n_bytes, int src_on_device);

/* ---- one process, several GPUs ("one process + N host threads"; BASELINE configs[4]) ---------------------------
 * A replica set = one handle per entry of device_ids, each holding a READ-ONLY copy of the owner's device map image, copied device
 * to device (hipMemcpyPeer; no host blob, no host mirror per replica -- FLS_REPLICAS_VIA_BLOB=1 restores round 3's fls_map_export once +
 * fls_map_import per device, which is also what the library falls back to, loudly, where a peer copy is refused); the owner itself
 * serves the first entry that names its own
 * device, so {owner's device} alone is valid and {d, d} gives two handles on one GPU (what the tests use on a one-GPU box).
 * fls_replicas_match_batch = fls_match_batch with the jobs block-partitioned over the entries (job j of the caller's arrays
 * lands on entry floor-partition(j); results are written straight into the caller's arrays: one address space, no gather),
 * one host thread per entry, `lanes` stream lanes per entry.  Every job still equals a fresh reference matcher holding the
 * owner's map, so the table is independent of the device list.  fls_replicas_refresh re-replicates after the owner's map
 * changed.  The owner must outlive the set and must not run a Match of its own while a batch is in flight.
 * FLS_P2PLANE_IVOX only (the kind with an exportable image); fls_replicas_import_ms: per entry, the last replication.      */
typedef struct fls_replicas* fls_replicas_handle;
fls_status fls_replicas_create(fls_handle owner, const int* device_ids, int n_devices, fls_replicas_handle* out);
fls_status fls_replicas_refresh(fls_replicas_handle r);
fls_status fls_replicas_match_batch(fls_replicas_handle r, size_t n_jobs, const float* const* src0, const size_t* n0, const float* const* src1,
                                    const size_t* n1, int stride_floats, double* T_colmajor, fls_stats* stats, int32_t* status, int lanes);
int fls_replicas_import_ms(fls_replicas_handle r, double* ms, int cap);
void fls_replicas_destroy(fls_replicas_handle r);

/* ---- VoxelGridCloud  (include/common/pointcloud_utility.h:216-271 = pcl::VoxelGrid<PointXYZI>::filter) -------------------------
 * The stand-alone filter the pipeline applies OUTSIDE the matchers: the planar / corner voxel filters of the preprocessing
 * thread that feed Match (src/slam/preprocessing.cpp:224-237) and the sub-map / multi-resolution filters of loop closure
 * (src/slam/loop_closure.cpp:215,251-252,258-259).  pts: n x stride floats (intensity as in fls_match); out: cap x 4 floats
 * {x, y, z, intensity}, one centroid per occupied leaf in ascending leaf index; *n_out = number of leaves (set even when out
 * is too small -> FLS_ERR_INVALID).
 *   FLS_VOXELGRID_EXACT   the reference's arithmetic bit for bit (leaf sums in libstdc++'s std::sort order) on the host worker pool;
 *                         needs no device.  PCL's "leaf size too small" case copies the input, as PCL does.
 *   FLS_VOXELGRID_DEVICE  on the GPU, the same bits (since round 4: the device sort reproduces libstdc++'s std::sort permutation,
 *                         csrc/kernels_exactsort.hpp; FLS_DEVICE_VOXELGRID=2 keeps the older ascending-point-index sums for A/B).
 *                         FLS_ERR_STATE when the device path declines (empty / no finite point / a non-finite point / the "leaf size
 *                         too small" case / n > 4,194,304 / introsort's heap-sort case on a long range): call again with FLS_VOXELGRID_EXACT.
 * "Bit for bit" is a statement about ONE toolchain: PCL 1.10 (Ubuntu 20.04 / ROS Noetic, what the reference's Dockerfile pulls) sorts the
 * leaf index vector with std::sort, and libstdc++'s introsort is what both filters reproduce.  PCL >= 1.11 sorts that vector with
 * boost::sort::spreadsort::integer_sort instead (unverified here: PCL is absent from this image): against such a build leaves of three or
 * more points can differ in the last float bits from EITHER filter -- the same order of magnitude as FLS_DEVICE_VOXELGRID=2. */
typedef enum fls_voxelgrid_mode { FLS_VOXELGRID_EXACT = 0, FLS_VOXELGRID_DEVICE = 1 } fls_voxelgrid_mode;
fls_status fls_voxel_grid_cloud(int device_id, fls_voxelgrid_mode mode, const float* pts, size_t n, int stride_floats, float leaf_size,
                                float* out, size_t cap_points, size_t* n_out);

/* ---- LoopClosure::Match  (src/slam/loop_closure.cpp:233-267) -----------------------------------------------------------------
 * The other registration consumer of the pipeline; it bypasses RegistrationInterface, so it gets its own entry point:
 *     float LoopClosure::Match(source_cloud, target_cloud, Mat4d& pose)
 * = pcl::NormalDistributionsTransform at resolutions 10 / 5 / 3 / 2 m (step size 0.5, 30 iterations) over
 * VoxelGridCloud(cloud, r * 0.2) clouds, then pcl::GeneralizedIterativeClosestPoint (30 iterations, 2.0 m correspondence
 * distance) over VoxelGridCloud(source, 0.5) / VoxelGridCloud(target, 0.4), return value gicp.getFitnessScore().
 * source / target: n x stride floats (as fls_match); T_colmajor: the initial guess in, the aligned pose out (target <- source);
 * *fitness: mean squared nearest-neighbour distance (FLT_MAX when GICP could not run: fewer than 20 filtered points).
 * Per-point work runs on the device, the six-parameter optimisers on the host (csrc/loop_closure.hpp).  No handle and no state a
 * caller can observe: the library keeps one set of device buffers per device for the life of the process (the second call on a
 * device allocates nothing); calls on one device run one after the other, calls on different devices side by side.       */
typedef struct fls_loop_stats {
    int32_t ndt_iterations[4];     /* per resolution stage: Newton iterations                       */
    int32_t ndt_evaluations[4];    /*                       score / derivative evaluations          */
    int32_t ndt_source_points[4];  /*                       source points after VoxelGridCloud      */
    int32_t ndt_target_leaves[4];  /*                       leaf Gaussians (>= 6 points)            */
    int32_t gicp_iterations, gicp_inner_iterations, gicp_evaluations, gicp_correspondences;
    int32_t gicp_source_points, gicp_target_points, gicp_failed, reserved;
    double ndt_score[4];           /* trans_probability_ of the stage                               */
    double T_after_ndt[16];        /* pose handed from the NDT stages to GICP                       */
} fls_loop_stats;
fls_status fls_loop_match(int device_id, const float* source, size_t n_source, const float* target, size_t n_target, int stride_floats,
                          double T_colmajor[16], float* fitness, fls_loop_stats* stats);

/* ---- RegistrationInterface::GetFitnessScore  (registration_interface.h:19) ---------------------- */
fls_status fls_get_fitness_score(fls_handle h, float max_range, float* score);

/* ---- resident-scan variant: the scan is uploaded once (and VoxelGrid-ed for ICP / NDT), then matched
 * from HBM.  fls_match == fls_scan_upload + fls_match_resident, for either value of update_map (the resident scan
 * keeps what a later map update needs until the next upload).  update_map == 0 skips the reference's whole
 * "converged && IsNeedAddCloud && !localization -> AddCloudToLocalMap" statement: such a call does not advance the
 * keyframe gate's last_T either.                                                                          */
fls_status fls_scan_upload(fls_handle h, const float* src0, size_t n0, const float* src1, size_t n1, int stride_floats);
fls_status fls_match_resident(fls_handle h, double T_colmajor[16], int update_map, fls_stats* stats);
/* fls_scan_upload_raw (revision 5): like fls_scan_upload, but for the kinds whose Match filters its source cloud first (IcpOptimized
 * icp_optimized.h:57, IncrementalNDT incremental_ndt.h:231-232) the RAW cloud stays resident and EVERY fls_match_resident that follows runs
 * that pcl::VoxelGrid itself before the iterations -- the whole of the reference's Match with its input already in device memory (what
 * bench.py reports for BASELINE configs[0] / [2]).  The other kinds: identical to fls_scan_upload.
 * Contract: the call withdraws the previously resident FILTERED scan at once -- between it and the next fls_match_resident the handle holds no
 * source cloud (fls_get_correspondences returns 0 rows, fls_get_fitness_score FLS_ERR_STATE until a Match has run on the new scan); it keeps one
 * packed host copy of the raw cloud (what the device filter declines goes to the host filter) and returns after the upload has completed. */
fls_status fls_scan_upload_raw(fls_handle h, const float* src0, size_t n0, const float* src1, size_t n1, int stride_floats);

/* ---- introspection (parity tests, DLOG-equivalent) ---------------------------------------------- */
/* pose / n_valid / sum_res after every executed iteration; returns the number of iterations logged. */
int fls_get_iteration_log(fls_handle h, double* T_iters /* cap x 16, col-major */, int32_t* n_valid, double* sum_res,
                          int cap);
/* neighbours held for each source point after the last Match: ids are map insertion ids (iVox), map cloud
 * indices (kd-tree kinds), voxel creation ids (NDT); K = 5 (1 for ICP, 7 for NDT).  slot 1 = corner set. */
int fls_get_correspondences(fls_handle h, int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap_points);
/* fls_map_size: slot 0 = map points (voxels for NDT), 1 = corner map (LoamFull), 102 = occupied voxels (iVox); slots >= 100 are
 * introspection counters of the device-side map update that the tests and bench.py read (device batches applied 103 / 109, refused
 * 104 / 110, voxels evicted inside device batches 117, of which re-created by a later point of the same batch 126, refusals by
 * reason 119-121, ...): see map_size() of the matcher in csrc/matcher_p2plane_ivox.hpp / matcher_ndt.hpp.  Not part of the drop-in. */
size_t fls_map_size(fls_handle h, int slot);

/* ---- measurement hooks ----------------------------------------------------------------------------
 * fls_set_profiling(h, flags): bit 0 = bracket every correspondence-kernel launch with hipEvents on the
 * handle's own stream; bit 1 = run the counting variant of the correspondence kernel, which tallies
 * probes / hit voxels / candidate points on the device (SURVEY.md 8d formula inputs; a few % slower).
 * fls_get_kernel_time returns the summed duration [ms] and launch count of the launches that did work
 * (early-exit launches after convergence are excluded) since the last call.                             */
fls_status fls_set_profiling(fls_handle h, int enable);
fls_status fls_get_kernel_time(fls_handle h, double* ms_total, int64_t* launches, uint64_t* point_iters);
/* traffic counters of the last Match run with flag bit 1 set */
fls_status fls_get_traffic_counters(fls_handle h, uint64_t* probes, uint64_t* hit_voxels, uint64_t* cand_points);

/* shader-clock stamps of the solve kernel's phases of the last iteration (only filled by -DFLS_TIMING builds) */
fls_status fls_get_debug_stamps(fls_handle h, int64_t out[16]);

/* test hook: the Gauss-Newton tail's 6x6 solver (Eigen FullPivHouseholderQR::solve semantics, one wave per system) on n
 * caller-supplied systems: H (n x 36, column-major), g (n x 6) -> x (n x 6).  tests/test_gpu_solver.py checks it bit for bit
 * against the oracle's restatement, rank-deficient systems included.                                                      */
fls_status fls_debug_fullpiv_qr6(int device_id, const double* H, const double* g, int n, double* x);

/* test hook: the SPD fast path in front of that solver (unpivoted LDL^T with the matrix rows in lanes 0..5 of one wave; csrc/wave_solve.hpp::
 * ldlt_solve6_wave) on n caller-supplied systems.  ok[s] = 1 where the fast path accepted system s (every pivot > 0, d_min > 1e-9 d_max, finite
 * solution) and x[s] is its solution; 0 where a Match would have gone on to fls_debug_fullpiv_qr6's solver (x[s] = 0).  tests/test_gpu_solver.py. */
fls_status fls_debug_ldlt6(int device_id, const double* H, const double* g, int n, double* x, int32_t* ok);

/* test hook (= fls_voxel_grid_cloud(..., FLS_VOXELGRID_DEVICE, ...)): the device VoxelGrid (pcl::VoxelGrid<PointXYZI>::filter semantics; the default source filter
 * of the ICP / NDT kinds, FLS_DEVICE_VOXELGRID=1) on a caller-supplied cloud: pts (n x stride floats, intensity as in fls_match) -> out (cap x 4
 * floats x, y, z, intensity), *n_out = number of leaves.  FLS_ERR_STATE when the device path declines (empty / no finite
 * point / PCL's "leaf size too small" case / n > 4,194,304: the matchers then run the host filter), FLS_ERR_INVALID when
 * out is too small (*n_out is still set).  Contract: csrc/kernels_voxelgrid.hpp; tests/test_gpu_voxelgrid.py.          */
fls_status fls_debug_voxel_grid(int device_id, const float* pts, size_t n, int stride, float leaf, float* out, size_t cap, size_t* n_out);
/* measurement hook (tools/gpu_vg_large.py): the same filter `reps` times on one resident cloud, wall-clock milliseconds of every repetition (the
 * call's own host waits included, upload excluded) in ms[0 .. reps); the first repetition allocates.                                      */
fls_status fls_debug_voxel_grid_timed(int device_id, const float* pts, size_t n, int stride, float leaf, int reps, double* ms, size_t* n_out);
/* Test hook of the device VoxelGrid's sort (csrc/kernels_exactsort.hpp): sorts the n records {key[i], val[i]} in place BY KEY ONLY, leaving
 * records of equal key in exactly the order libstdc++'s std::sort leaves them in (what pcl::VoxelGrid's leaf sums depend on,
 * include/common/pointcloud_utility.h:216-271).  on_host = 1: std::sort on the host (the reference permutation, no GPU needed);
 * on_host = 0: the device kernels, host-steered sequence; on_host = 2: the launch sequence the one-stream VoxelGrid queues (pre-enqueued
 * levels + the task kernel, no host round trip).  FLS_ERR_STATE: the device declined (a range still longer than 2,048 records when introsort's depth
 * limit is reached -- the heap-sort case of shorter ranges is reproduced on the device -- or more than 4 Mi records).
 * fls_debug_exact_sort_marks: progress stamps of the sort kernel last run with FLS_ES_DEBUG set (diagnostics, tools/es_marks.py). */
fls_status fls_debug_exact_sort(int device_id, uint32_t* key, uint32_t* val, size_t n, int on_host);
int fls_debug_exact_sort_marks(unsigned* out, int n);

const char* fls_status_string(int status);
int fls_abi_version(void);
int fls_abi_revision(void);
/* number of visible HIP devices whose arch is gfx950 (0 => every compute call fails with FLS_ERR_DEVICE) */
int fls_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* FLS_REG_H */
