"""Python mirror of the reference's registration plug-in interface, backed by libfls_reg.so.

The production integration is C++ (include/fls_hip_registration.h implements the
reference's ``RegistrationInterface`` on top of the same C ABI); this module gives the
tests and bench.py the same surface with the reference's class names, constructor
argument order and semantics:

    RegistrationInterface            include/registration/registration_interface.h:11-20
      Match(cluster, T) -> bool      (T is in/out: a (4,4) float64 numpy array, world <- body)
      AddCloudToLocalMap([clouds])
      GetFitnessScore(max_range)
    PointcloudCluster                include/lidar/pointcloud_cluster.h:13-85 (the three clouds Match reads)

Clouds are (n,3|4|8) float32 arrays (xyz[i][pad], the 8-wide form is pcl::PointXYZI's memory).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import FlsError, Params, Stats

FloatNaN = float(np.finfo(np.float32).max)  # include/common/constant_variable.h:11

# mode strings added next to include/common/constant_variable.h:21-25 by the integration
kPointToPlane_IVOX_HIP = "PointToPlane_IVOX_HIP"
kPointToPlane_KdTree_HIP = "PointToPlane_KdTree_HIP"
kLoamFull_KdTree_HIP = "LoamFull_KdTree_HIP"
kIcpOptimized_HIP = "IcpOptimized_HIP"
kIncrementalNDT_HIP = "IncrementalNDT_HIP"


@dataclass
class PointcloudCluster:
    ordered_cloud_: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    planar_cloud_: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    corner_cloud_: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    # LOAM front-end fields (lidar/pointcloud_cluster.h): raw driver cloud in, range-image bookkeeping out
    raw_cloud_: np.ndarray = None          # structured array with x, y, z, intensity, ring (e.g. synth.RAW_POINT_DTYPE)
    point_depth_vec_: np.ndarray = None
    point_col_index_vec_: np.ndarray = None
    row_start_index_vec_: np.ndarray = None
    row_end_index_vec_: np.ndarray = None
    feature_state_: object = None          # device-resident projection the FeatureExtractor continues from


def _cloud(c):
    if c is None:
        return None, None, 0, 3
    a = np.ascontiguousarray(c, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be (n, >=3) float32")
    return a, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]


class RegistrationInterface:
    """Base: owns one fls_handle.  Not thread-safe (one owner thread, like the reference)."""

    KIND = None
    K = 5

    def __init__(self, params: Params, device_id: int = 0):
        self.params = params
        self.stats = Stats()
        self._h = C.c_void_p()
        rc = _lib.lib().fls_create(self.KIND, C.byref(params), device_id, C.byref(self._h))
        if rc != _lib.FLS_OK:
            self._h = C.c_void_p()
            raise FlsError(rc, f"fls_create(kind={self.KIND})")

    # -- the three virtuals -------------------------------------------------------------------
    def AddCloudToLocalMap(self, cloud_list) -> None:
        clouds = list(cloud_list)
        a0, p0, n0, s0 = _cloud(clouds[0])
        a1, p1, n1, s1 = _cloud(clouds[1]) if len(clouds) > 1 else (None, None, 0, s0)
        if a1 is not None and s1 != s0:
            raise ValueError("clouds must share a stride")
        rc = _lib.lib().fls_add_cloud_to_local_map(self._h, p0, n0, p1, n1, s0)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_add_cloud_to_local_map")

    def _sources(self, cluster: PointcloudCluster):
        return cluster.planar_cloud_, None

    def Match(self, cluster: PointcloudCluster, T: np.ndarray, update_map: bool = True) -> bool:
        s0, s1 = self._sources(cluster)
        a0, p0, n0, st0 = _cloud(s0)
        a1, p1, n1, st1 = _cloud(s1)
        if a1 is not None and st1 != st0:
            raise ValueError("clouds must share a stride")
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).reshape(-1)  # column-major
        rc = _lib.lib().fls_match(self._h, p0, n0, p1, n1, st0, Tf.ctypes.data_as(C.POINTER(C.c_double)),
                                  1 if update_map else 0, C.byref(self.stats))
        if rc < 0:
            raise FlsError(rc, "fls_match")
        T[...] = Tf.reshape(4, 4).T
        return rc == _lib.FLS_OK

    def _batch_args(self, clusters, T_inits):
        n = len(clusters)
        fp = C.POINTER(C.c_float)
        keep, p0s, n0s, p1s, n1s = [], (fp * n)(), (C.c_size_t * n)(), (fp * n)(), (C.c_size_t * n)()
        stride, two = None, False
        for j, cl in enumerate(clusters):
            s0, s1 = self._sources(cl)
            a0, p0, n0, st0 = _cloud(s0)
            a1, p1, n1, st1 = _cloud(s1)
            if stride not in (None, st0) or (a1 is not None and st1 != st0):
                raise ValueError("clouds must share a stride")
            stride = st0
            keep += [a0, a1]
            p0s[j], n0s[j] = p0, n0
            if a1 is not None:
                two = True
                p1s[j], n1s[j] = p1, n1
        Tf = np.ascontiguousarray(np.asarray(T_inits, dtype=np.float64).reshape(n, 4, 4).transpose(0, 2, 1)).reshape(-1)  # column-major
        return n, keep, p0s, n0s, (p1s if two else None), (n1s if two else None), stride or 3, Tf, (Stats * n)(), (C.c_int32 * n)()

    def MatchBatch(self, clusters, T_inits, lanes: int = 4):
        """fls_match_batch: independent registrations of `clusters[j]` from `T_inits[j]` against the current map (no map
        update, no state carried between jobs).  Returns (ok[j], T[j] (n,4,4), stats[j]) -- BASELINE configs[4]."""
        n, keep, p0s, n0s, p1s, n1s, stride, Tf, stats, status = self._batch_args(clusters, T_inits)
        rc = _lib.lib().fls_match_batch(self._h, n, p0s, n0s, p1s, n1s, stride, Tf.ctypes.data_as(C.POINTER(C.c_double)), stats, status, int(lanes))
        if rc < 0:
            raise FlsError(rc, "fls_match_batch")
        T = Tf.reshape(n, 4, 4).transpose(0, 2, 1).copy()
        return [status[j] == _lib.FLS_OK for j in range(n)], T, list(stats)

    def Replicas(self, device_ids):
        """fls_replicas_create: one handle per entry of device_ids, each with a copy of this handle's map (one process, N GPUs)."""
        return ReplicaSet(self, device_ids)

    def GetFitnessScore(self, max_range: float) -> float:
        out = C.c_float()
        rc = _lib.lib().fls_get_fitness_score(self._h, max_range, C.byref(out))
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_get_fitness_score")
        return float(out.value)

    # -- resident-scan variant (benchmarks) -------------------------------------------------------
    def UploadScan(self, cluster: PointcloudCluster) -> None:
        s0, s1 = self._sources(cluster)
        a0, p0, n0, st0 = _cloud(s0)
        a1, p1, n1, st1 = _cloud(s1)
        rc = _lib.lib().fls_scan_upload(self._h, p0, n0, p1, n1, st0)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_scan_upload")

    def UploadScanRaw(self, cluster: PointcloudCluster) -> None:
        """fls_scan_upload_raw: the raw cloud stays resident; every MatchResident runs the source VoxelGrid itself (ICP / NDT)."""
        s0, s1 = self._sources(cluster)
        a0, p0, n0, st0 = _cloud(s0)
        a1, p1, n1, st1 = _cloud(s1)
        rc = _lib.lib().fls_scan_upload_raw(self._h, p0, n0, p1, n1, st0)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_scan_upload_raw")

    def MatchResident(self, T: np.ndarray, update_map: bool = False) -> bool:
        # hot in bench.py: one preallocated column-major pose buffer, no per-call ctypes object construction
        buf = getattr(self, "_Tbuf", None)
        if buf is None:
            buf = self._Tbuf = np.zeros(16, dtype=np.float64)
            self._Tptr = buf.ctypes.data_as(C.POINTER(C.c_double))
            self._Tview = buf.reshape(4, 4)  # row r of the view = column r of the pose
            self._stats_ref = C.byref(self.stats)
            self._fn_resident = _lib.lib().fls_match_resident
        self._Tview[...] = T.T
        rc = self._fn_resident(self._h, self._Tptr, 1 if update_map else 0, self._stats_ref)
        if rc < 0:
            raise FlsError(rc, "fls_match_resident")
        T[...] = self._Tview.T
        return rc == _lib.FLS_OK

    def resident_call(self, T_init: np.ndarray, update_map: bool = False):
        """Pre-bound zero-overhead form of MatchResident for tight loops: returns (run, T_view) where run()
        re-registers from T_init (copied into the column-major call buffer by memmove) and returns the status;
        T_view is the (4,4) row-major-looking view of the result (transpose of the column-major buffer)."""
        init = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64).reshape(4, 4).T).reshape(-1).copy()
        buf = np.zeros(16, dtype=np.float64)
        ptr = buf.ctypes.data_as(C.POINTER(C.c_double))
        fn, h, flag, st = _lib.lib().fls_match_resident, self._h, 1 if update_map else 0, C.byref(self.stats)
        src, dst, nbytes = init.ctypes.data, buf.ctypes.data, 128
        move = C.memmove

        def run():
            move(dst, src, nbytes)
            return fn(h, ptr, flag, st)

        self._keep = (init, buf)
        return run, buf.reshape(4, 4).T

    # -- introspection --------------------------------------------------------------------------
    def iteration_log(self, cap: int = 64):
        T = np.zeros((cap, 16)); nv = np.zeros(cap, np.int32); sr = np.zeros(cap)
        n = _lib.lib().fls_get_iteration_log(self._h, T.ctypes.data_as(C.POINTER(C.c_double)),
                                             nv.ctypes.data_as(C.POINTER(C.c_int32)), sr.ctypes.data_as(C.POINTER(C.c_double)), cap)
        n = min(n, cap)
        Ts = np.stack([T[i].reshape(4, 4).T for i in range(n)]) if n else np.zeros((0, 4, 4))
        return Ts, nv[:n].copy(), sr[:n].copy()

    def correspondences(self, slot: int = 0):
        n = self.stats.n_source_corner if slot == 1 else self.stats.n_source
        ids = np.full((n, self.K), -1, np.int32); cnt = np.zeros(n, np.uint8); valid = np.zeros(n, np.uint8)
        if n:
            m = _lib.lib().fls_get_correspondences(self._h, slot, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   cnt.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                   valid.ctypes.data_as(C.POINTER(C.c_uint8)), n)
            if m < 0:
                raise FlsError(_lib.FLS_ERR_DEVICE, "fls_get_correspondences")
        return ids, cnt, valid

    def ExportMap(self) -> np.ndarray:
        """fls_map_export: the map image as a self-contained byte blob (uint8 array), e.g. to broadcast it to other GPUs."""
        n = _lib.lib().fls_map_export(self._h, None, 0)
        if n == 0:
            raise FlsError(_lib.FLS_ERR_STATE, "fls_map_export (this kind has no exportable image)")
        blob = np.empty(n, np.uint8)
        m = _lib.lib().fls_map_export(self._h, blob.ctypes.data, n)
        if m != n:
            raise FlsError(_lib.FLS_ERR_STATE, "fls_map_export")
        return blob

    def ImportMap(self, blob: np.ndarray) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        rc = _lib.lib().fls_map_import(self._h, blob.ctypes.data, blob.size)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_map_import")

    def MapImageBytes(self) -> int:
        """fls_map_image_bytes: size of the flat DEVICE image (0: this kind has none)."""
        return int(_lib.lib().fls_map_image_bytes(self._h))

    def ExportMapImage(self, ptr: int, nbytes: int, on_device: bool) -> None:
        """fls_map_image_export into caller memory (`ptr`: device memory of this handle's device, or host memory)."""
        rc = _lib.lib().fls_map_image_export(self._h, C.c_void_p(ptr), nbytes, 1 if on_device else 0)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_map_image_export")

    def ImportMapImage(self, ptr: int, nbytes: int, on_device: bool) -> None:
        """fls_map_image_import: this handle becomes a read-only replica of the exporter's map."""
        rc = _lib.lib().fls_map_image_import(self._h, C.c_void_p(ptr), nbytes, 1 if on_device else 0)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_map_image_import")

    def map_size(self, slot: int = 0) -> int:
        return int(_lib.lib().fls_map_size(self._h, slot))

    def set_profiling(self, events: bool, counters: bool = False) -> None:
        _lib.lib().fls_set_profiling(self._h, (1 if events else 0) | (2 if counters else 0))

    def kernel_time(self):
        ms = C.c_double(); n = C.c_int64(); pi = C.c_uint64()
        _lib.lib().fls_get_kernel_time(self._h, C.byref(ms), C.byref(n), C.byref(pi))
        return ms.value, n.value, pi.value

    def traffic_counters(self):
        a = C.c_uint64(); b = C.c_uint64(); c = C.c_uint64()
        _lib.lib().fls_get_traffic_counters(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().fls_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoamPointToPlaneIVOX(RegistrationInterface):
    """loam_point_to_plane_ivox.h:37-51 ctor argument order."""

    KIND = _lib.P2PLANE_IVOX

    def __init__(self, point_to_planar_thres, position_converge_thres, rotation_converge_thres, opti_iter_num=30,
                 is_localization_mode=False, device_id=0):
        super().__init__(Params(point_to_planar_thres=point_to_planar_thres, position_converge_thres=position_converge_thres,
                                rotation_converge_thres=rotation_converge_thres, max_iterations=opti_iter_num,
                                is_localization_mode=int(is_localization_mode)), device_id)


class IcpOptimized(RegistrationInterface):
    """icp_optimized.h:23-32 ctor argument order."""

    KIND = _lib.ICP_OPTIMIZED
    K = 1

    def __init__(self, max_iterations, local_map_size, map_cloud_filter_size, source_cloud_filter_size, max_correspond_distance,
                 position_converge_thres, rotation_converge_thres, rot_thre_add_cloud, dist_thre_add_cloud,
                 is_localization_mode=False, device_id=0):
        super().__init__(Params(max_iterations=max_iterations, local_map_size=local_map_size,
                                map_cloud_filter_size=map_cloud_filter_size, source_cloud_filter_size=source_cloud_filter_size,
                                point_search_thres=max_correspond_distance, position_converge_thres=position_converge_thres,
                                rotation_converge_thres=rotation_converge_thres, rot_thre_add_cloud=rot_thre_add_cloud,
                                dist_thre_add_cloud=dist_thre_add_cloud, is_localization_mode=int(is_localization_mode)), device_id)

    def _sources(self, cluster):
        return cluster.ordered_cloud_, None


class IncrementalNDT(RegistrationInterface):
    """incremental_ndt.h:22-26 ctor argument order."""

    KIND = _lib.INCREMENTAL_NDT
    K = 7

    def __init__(self, voxel_size, res_outlier_threshold, source_cloud_filter_size, rotation_converge_thres, position_converge_thres,
                 min_points_in_voxel, max_points_in_voxel, min_effective_pts, capacity, max_iteration, is_localization_mode=False,
                 device_id=0):
        super().__init__(Params(ndt_voxel_size=voxel_size, ndt_res_outlier_threshold=res_outlier_threshold,
                                source_cloud_filter_size=source_cloud_filter_size, rotation_converge_thres=rotation_converge_thres,
                                position_converge_thres=position_converge_thres, ndt_min_points_in_voxel=min_points_in_voxel,
                                ndt_max_points_in_voxel=max_points_in_voxel, ndt_min_effective_pts=min_effective_pts,
                                ndt_capacity=capacity, max_iterations=max_iteration, is_localization_mode=int(is_localization_mode)),
                         device_id)

    def _sources(self, cluster):
        return cluster.ordered_cloud_, None


class LoamFull(RegistrationInterface):
    """loam_full_kdtree.h:31-36 ctor argument order.  AddCloudToLocalMap([planar, corner])."""

    KIND = _lib.LOAM_FULL

    def __init__(self, point_to_planar_thres, point_search_thres, line_ratio_thres, position_converge_thres, rotation_converge_thres,
                 dist_thre_add_cloud, rot_thre_add_cloud, local_corner_size, local_planar_size, corner_voxel_filter_size,
                 planar_voxel_filter_size, max_iteration, device_id=0):
        super().__init__(Params(point_to_planar_thres=point_to_planar_thres, point_search_thres=point_search_thres,
                                line_ratio_thres=line_ratio_thres, position_converge_thres=position_converge_thres,
                                rotation_converge_thres=rotation_converge_thres, dist_thre_add_cloud=dist_thre_add_cloud,
                                rot_thre_add_cloud=rot_thre_add_cloud, local_corner_size=local_corner_size,
                                local_planar_size=local_planar_size, corner_voxel_filter_size=corner_voxel_filter_size,
                                planar_voxel_filter_size=planar_voxel_filter_size, max_iterations=max_iteration), device_id)

    def _sources(self, cluster):
        return cluster.planar_cloud_, cluster.corner_cloud_


class LoamPointToPlaneKdtree(RegistrationInterface):
    """loam_point_to_plane_kdtree.h:32-42 ctor argument order."""

    KIND = _lib.P2PLANE_KDTREE

    def __init__(self, point_to_planar_thres, position_converge_thres, rotation_converge_thres, rot_thre_add_cloud,
                 dist_thre_add_cloud, local_map_size, map_cloud_filter_size, opti_iter_num=30, is_localization_mode=False,
                 device_id=0):
        super().__init__(Params(point_to_planar_thres=point_to_planar_thres, position_converge_thres=position_converge_thres,
                                rotation_converge_thres=rotation_converge_thres, rot_thre_add_cloud=rot_thre_add_cloud,
                                dist_thre_add_cloud=dist_thre_add_cloud, local_map_size=local_map_size,
                                map_cloud_filter_size=map_cloud_filter_size, max_iterations=opti_iter_num,
                                is_localization_mode=int(is_localization_mode)), device_id)


class LoopStats(C.Structure):
    """fls_loop_stats (include/fls_reg.h)"""
    _fields_ = [("ndt_iterations", C.c_int32 * 4), ("ndt_evaluations", C.c_int32 * 4), ("ndt_source_points", C.c_int32 * 4), ("ndt_target_leaves", C.c_int32 * 4),
                ("gicp_iterations", C.c_int32), ("gicp_inner_iterations", C.c_int32), ("gicp_evaluations", C.c_int32), ("gicp_correspondences", C.c_int32),
                ("gicp_source_points", C.c_int32), ("gicp_target_points", C.c_int32), ("gicp_failed", C.c_int32), ("reserved", C.c_int32),
                ("ndt_score", C.c_double * 4), ("T_after_ndt", C.c_double * 16)]


def LoopClosureMatch(source_cloud: np.ndarray, target_cloud: np.ndarray, pose: np.ndarray, device_id: int = 0):
    """LoopClosure::Match (src/slam/loop_closure.cpp:233-267): 4-resolution NDT + GICP.  `pose` (4,4) is in/out like the reference's
    Mat4d&; returns (fitness, LoopStats)."""
    a, pa, na, sa = _cloud(source_cloud)
    b, pb, nb, sb = _cloud(target_cloud)
    if sa != sb:
        raise ValueError("clouds must share a stride")
    Tf = np.ascontiguousarray(np.asarray(pose, dtype=np.float64).reshape(4, 4).T).reshape(-1).copy()
    fit = C.c_float()
    st = LoopStats()
    rc = _lib.lib().fls_loop_match(device_id, pa, na, pb, nb, sa, Tf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(fit), C.byref(st))
    if rc != _lib.FLS_OK:
        raise FlsError(rc, "fls_loop_match")
    pose[...] = Tf.reshape(4, 4).T
    return float(fit.value), st


def VoxelGridCloud(cloud: np.ndarray, voxel_size: float = 0.1, on_device: bool = False, device_id: int = 0) -> np.ndarray:
    """include/common/pointcloud_utility.h:216-271 (pcl::VoxelGrid<PointXYZI>::filter): the stand-alone filter of the preprocessing
    thread (preprocessing.cpp:224-237) and of loop closure.  (n, 3|4|8) float32 -> (m, 4) {x, y, z, intensity}.
    on_device = False: the reference's arithmetic bit for bit on the host worker pool (needs no GPU);
    on_device = True:  the device filter (contract: csrc/kernels_voxelgrid.hpp); falls back to the exact filter when it declines."""
    a, ptr, n, stride = _cloud(cloud)
    out = np.zeros((max(n, 1), 4), np.float32)
    n_out = C.c_size_t(0)
    fp = C.POINTER(C.c_float)
    L = _lib.lib()
    rc = _lib.FLS_ERR_STATE
    if on_device:
        rc = L.fls_voxel_grid_cloud(device_id, 1, ptr, n, stride, np.float32(voxel_size), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
        if rc not in (_lib.FLS_OK, _lib.FLS_ERR_STATE):
            raise FlsError(rc, "fls_voxel_grid_cloud(device)")
    if rc != _lib.FLS_OK:
        rc = L.fls_voxel_grid_cloud(device_id, 0, ptr, n, stride, np.float32(voxel_size), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_voxel_grid_cloud")
    return out[: n_out.value].copy()


def make_matcher(mode: str, cfg: dict, is_localization_mode: bool = False, device_id: int = 0) -> RegistrationInterface:
    """FrontEnd::InitMatcher / Localization::InitMatcher (src/slam/frontend.cpp:30-88, localization.cpp:43-92):
    choose the implementation from the YAML mode string, arguments from the `registration` YAML block."""
    r = cfg
    if mode in ("PointToPlane_IVOX", kPointToPlane_IVOX_HIP):
        return LoamPointToPlaneIVOX(r["point_to_planar_thres"], r["position_converge_thres"], r["rotation_converge_thres"],
                                    r["optimization_iter_num"], is_localization_mode, device_id)
    if mode in ("IcpOptimized", kIcpOptimized_HIP):
        return IcpOptimized(r["optimization_iter_num"], r["local_map_size"], r["local_map_cloud_filter_size"],
                            r["source_cloud_filter_size"], r["point_search_thres"], r["position_converge_thres"],
                            r["rotation_converge_thres"], r["keyframe_delta_rotation"], r["keyframe_delta_distance"],
                            is_localization_mode, device_id)
    if mode in ("IncrementalNDT", kIncrementalNDT_HIP):
        return IncrementalNDT(r["ndt_voxel_size"], r["ndt_outlier_threshold"], r["source_cloud_filter_size"],
                              r["rotation_converge_thres"], r["position_converge_thres"], r["ndt_min_points_in_voxel"],
                              r["ndt_max_points_in_voxel"], r["ndt_min_effective_pts"], r["ndt_capacity"], r["optimization_iter_num"],
                              is_localization_mode, device_id)
    if mode in ("LoamFull_KdTree", kLoamFull_KdTree_HIP):
        return LoamFull(r["point_to_planar_thres"], r["point_search_thres"], r["line_ratio_thres"], r["position_converge_thres"],
                        r["rotation_converge_thres"], r["keyframe_delta_distance"], r["keyframe_delta_rotation"],
                        r["local_corner_map_size"], r["local_planar_map_size"], r["local_corner_voxel_filter_size"],
                        r["local_planar_voxel_filter_size"], r["optimization_iter_num"], device_id)
    if mode in ("PointToPlane_KdTree", kPointToPlane_KdTree_HIP):
        return LoamPointToPlaneKdtree(r["point_to_planar_thres"], r["position_converge_thres"], r["rotation_converge_thres"],
                                      r["keyframe_delta_rotation"], r["keyframe_delta_distance"], r["local_map_size"],
                                      r["local_map_cloud_filter_size"], r["optimization_iter_num"], is_localization_mode, device_id)
    raise ValueError(f"Unsupported registration type {mode!r}")


# the reference's YAML `registration:` blocks
YAML_NCLT_IVOX = dict(optimization_iter_num=10, point_to_planar_thres=0.1, position_converge_thres=0.005,
                      rotation_converge_thres=0.001)  # config/mapping/config_nclt.yaml:43-47
YAML_NCLT_ICP = dict(local_map_size=50, point_search_thres=1.0, optimization_iter_num=30, local_map_cloud_filter_size=0.4,
                     source_cloud_filter_size=0.4, position_converge_thres=0.005, rotation_converge_thres=0.005,
                     keyframe_delta_distance=1.0, keyframe_delta_rotation=0.2)  # config_nclt_icp.yaml:41-50
YAML_NCLT_NDT = dict(ndt_voxel_size=1.0, ndt_outlier_threshold=5.0, source_cloud_filter_size=0.2, optimization_iter_num=30,
                     ndt_min_points_in_voxel=5, ndt_max_points_in_voxel=50, ndt_min_effective_pts=50, ndt_capacity=100000,
                     position_converge_thres=0.005, rotation_converge_thres=0.005)  # config_nclt_ndt.yaml:41-51
YAML_NCLT_LOAM_FULL = dict(optimization_iter_num=30, local_corner_map_size=50, local_planar_map_size=40, point_search_thres=1.0,
                           line_ratio_thres=3.0, point_to_planar_thres=0.2, position_converge_thres=0.01,
                           rotation_converge_thres=0.05, keyframe_delta_distance=1.0, keyframe_delta_rotation=0.2,
                           local_corner_voxel_filter_size=0.2, local_planar_voxel_filter_size=0.4)  # config_nclt_loam_full.yaml:47-58
YAML_NCLT_LOC_KDTREE = dict(optimization_iter_num=8, point_to_planar_thres=0.1, position_converge_thres=0.005,
                            rotation_converge_thres=0.005, local_map_size=0, local_map_cloud_filter_size=0.5,
                            keyframe_delta_distance=0.0, keyframe_delta_rotation=0.0)  # config/localization/config_nclt.yaml:40-50


class ReplicaSet:
    """Native one-process multi-GPU form of BASELINE configs[4] (include/fls_reg.h: fls_replicas_*): the owner's map image
    replicated per device, jobs block-partitioned over the devices, one host thread per device inside the library."""

    def __init__(self, owner, device_ids):
        self._owner = owner  # keeps the owner handle alive
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        rc = _lib.lib().fls_replicas_create(owner._h, ids, len(device_ids), C.byref(h))
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_replicas_create")
        self._r = h
        self.n = len(device_ids)

    def close(self):
        if getattr(self, "_r", None):
            _lib.lib().fls_replicas_destroy(self._r)
            self._r = None

    __del__ = close

    def Refresh(self):
        rc = _lib.lib().fls_replicas_refresh(self._r)
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_replicas_refresh")

    def import_ms(self):
        buf = (C.c_double * self.n)()
        _lib.lib().fls_replicas_import_ms(self._r, buf, self.n)
        return list(buf)

    def MatchBatch(self, clusters, T_inits, lanes: int = 4):
        n, keep, p0s, n0s, p1s, n1s, stride, Tf, stats, status = self._owner._batch_args(clusters, T_inits)
        rc = _lib.lib().fls_replicas_match_batch(self._r, n, p0s, n0s, p1s, n1s, stride, Tf.ctypes.data_as(C.POINTER(C.c_double)), stats, status, int(lanes))
        if rc < 0:
            raise FlsError(rc, "fls_replicas_match_batch")
        T = Tf.reshape(n, 4, 4).transpose(0, 2, 1).copy()
        return [status[j] == _lib.FLS_OK for j in range(n)], T, list(stats)
