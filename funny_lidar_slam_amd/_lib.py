"""ctypes binding of libfls_reg.so (the product's C ABI, include/fls_reg.h).

No fallback of any kind: if the shared library is missing, or no gfx950 device
is usable, the calls fail loudly (RuntimeError / FlsError).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLS_REG_LIB") or os.path.join(_HERE, "libfls_reg.so")  # FLS_REG_LIB: diagnosis builds only (csrc/Makefile `timing`)
CSRC_DIR = os.path.join(_HERE, "csrc")

# every symbol include/fls_reg.h declares (tests check they are all exported)
EXPORTED_SYMBOLS = [
    "fls_create", "fls_destroy", "fls_add_cloud_to_local_map", "fls_match", "fls_get_fitness_score",
    "fls_scan_upload", "fls_scan_upload_raw", "fls_match_resident", "fls_match_batch", "fls_map_export", "fls_map_import", "fls_map_image_bytes", "fls_map_image_export", "fls_map_image_import", "fls_get_iteration_log", "fls_get_correspondences", "fls_map_size",
    "fls_set_profiling", "fls_get_kernel_time", "fls_get_traffic_counters", "fls_get_debug_stamps", "fls_debug_fullpiv_qr6", "fls_debug_ldlt6", "fls_debug_voxel_grid", "fls_debug_voxel_grid_timed", "fls_debug_exact_sort", "fls_debug_exact_sort_marks", "fls_voxel_grid_cloud", "fls_loop_match", "fls_status_string", "fls_abi_version", "fls_abi_revision",
    "fls_device_count",
    "fls_replicas_create", "fls_replicas_refresh", "fls_replicas_match_batch", "fls_replicas_import_ms", "fls_replicas_destroy",
    # include/fls_features.h
    "fls_features_create", "fls_features_destroy", "fls_features_project", "fls_features_extract", "fls_features_get", "fls_features_get_time",
]

FLS_OK, FLS_NOT_CONVERGED, FLS_SKIPPED = 0, 1, 2
FLS_ERR_INVALID, FLS_ERR_DEVICE, FLS_ERR_RANGE, FLS_ERR_NOMEM, FLS_ERR_STATE = -1, -2, -3, -4, -5

ICP_OPTIMIZED, P2PLANE_IVOX, INCREMENTAL_NDT, LOAM_FULL, P2PLANE_KDTREE = range(5)


class FlsError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        super().__init__(f"{where}: {status_string(status)} ({status})")


class Params(C.Structure):
    """fls_params (include/fls_reg.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("max_iterations", C.c_uint32),
        ("is_localization_mode", C.c_int32),
        ("local_map_size", C.c_uint32),
        ("local_corner_size", C.c_uint32),
        ("local_planar_size", C.c_uint32),
        ("ndt_min_points_in_voxel", C.c_int32),
        ("ndt_max_points_in_voxel", C.c_int32),
        ("ndt_min_effective_pts", C.c_int32),
        ("ndt_capacity", C.c_int32),
        ("map_cloud_filter_size", C.c_float),
        ("source_cloud_filter_size", C.c_float),
        ("corner_voxel_filter_size", C.c_float),
        ("planar_voxel_filter_size", C.c_float),
        ("point_to_planar_thres", C.c_double),
        ("point_search_thres", C.c_double),
        ("line_ratio_thres", C.c_double),
        ("position_converge_thres", C.c_double),
        ("rotation_converge_thres", C.c_double),
        ("rot_thre_add_cloud", C.c_double),
        ("dist_thre_add_cloud", C.c_double),
        ("ndt_voxel_size", C.c_double),
        ("ndt_res_outlier_threshold", C.c_double),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(Params)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class Stats(C.Structure):
    """fls_stats (include/fls_reg.h)."""

    _fields_ = [
        ("iterations", C.c_int32),
        ("converged", C.c_int32),
        ("n_valid", C.c_int32),
        ("n_valid_corner", C.c_int32),
        ("n_source", C.c_int32),
        ("n_source_corner", C.c_int32),
        ("map_updated", C.c_int32),
        ("reserved", C.c_int32),
        ("sum_res", C.c_double),
        ("sum_res_corner", C.c_double),
        ("last_dx", C.c_double * 6),
    ]


class FeatureParams(C.Structure):
    """fls_feature_params (include/fls_features.h)."""

    _fields_ = [("struct_size", C.c_uint32), ("lidar_vertical_scan", C.c_int32), ("lidar_horizontal_scan", C.c_int32),
                ("lidar_horizontal_resolution", C.c_float), ("min_distance", C.c_float), ("max_distance", C.c_float),
                ("corner_thres", C.c_float), ("planar_thres", C.c_float), ("corner_voxel_filter_size", C.c_float),
                ("planar_voxel_filter_size", C.c_float)]


class PointLayout(C.Structure):
    """fls_point_layout (include/fls_features.h)."""

    _fields_ = [("stride_bytes", C.c_uint32), ("xyz_offset", C.c_uint32), ("intensity_offset", C.c_uint32), ("ring_offset", C.c_uint32)]


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libfls_reg.so (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC_DIR, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the registration hot path)")
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        bp = C.POINTER(C.c_uint8)
        hp = C.c_void_p
        L.fls_create.restype = C.c_int
        L.fls_create.argtypes = [C.c_int, C.POINTER(Params), C.c_int, C.POINTER(hp)]
        L.fls_destroy.restype = None
        L.fls_destroy.argtypes = [hp]
        L.fls_add_cloud_to_local_map.restype = C.c_int
        L.fls_add_cloud_to_local_map.argtypes = [hp, fp, C.c_size_t, fp, C.c_size_t, C.c_int]
        L.fls_match.restype = C.c_int
        L.fls_match.argtypes = [hp, fp, C.c_size_t, fp, C.c_size_t, C.c_int, dp, C.c_int, C.POINTER(Stats)]
        L.fls_get_fitness_score.restype = C.c_int
        L.fls_get_fitness_score.argtypes = [hp, C.c_float, fp]
        L.fls_scan_upload.restype = C.c_int
        L.fls_scan_upload.argtypes = [hp, fp, C.c_size_t, fp, C.c_size_t, C.c_int]
        L.fls_map_image_bytes.restype = C.c_size_t
        L.fls_map_image_bytes.argtypes = [hp]
        L.fls_map_image_export.restype = C.c_int
        L.fls_map_image_export.argtypes = [hp, C.c_void_p, C.c_size_t, C.c_int]
        L.fls_map_image_import.restype = C.c_int
        L.fls_map_image_import.argtypes = [hp, C.c_void_p, C.c_size_t, C.c_int]
        L.fls_scan_upload_raw.restype = C.c_int
        L.fls_scan_upload_raw.argtypes = [hp, fp, C.c_size_t, fp, C.c_size_t, C.c_int]
        L.fls_match_resident.restype = C.c_int
        L.fls_match_resident.argtypes = [hp, dp, C.c_int, C.POINTER(Stats)]
        L.fls_match_batch.restype = C.c_int
        L.fls_match_batch.argtypes = [hp, C.c_size_t, C.POINTER(fp), C.POINTER(C.c_size_t), C.POINTER(fp), C.POINTER(C.c_size_t), C.c_int,
                                      dp, C.POINTER(Stats), ip, C.c_int]
        L.fls_replicas_create.restype = C.c_int
        L.fls_replicas_create.argtypes = [hp, ip, C.c_int, C.POINTER(C.c_void_p)]
        L.fls_replicas_refresh.restype = C.c_int
        L.fls_replicas_refresh.argtypes = [C.c_void_p]
        L.fls_replicas_match_batch.restype = C.c_int
        L.fls_replicas_match_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(fp), C.POINTER(C.c_size_t), C.POINTER(fp), C.POINTER(C.c_size_t), C.c_int,
                                               dp, C.POINTER(Stats), ip, C.c_int]
        L.fls_replicas_import_ms.restype = C.c_int
        L.fls_replicas_import_ms.argtypes = [C.c_void_p, dp, C.c_int]
        L.fls_replicas_destroy.restype = None
        L.fls_replicas_destroy.argtypes = [C.c_void_p]
        L.fls_get_iteration_log.restype = C.c_int
        L.fls_get_iteration_log.argtypes = [hp, dp, ip, dp, C.c_int]
        L.fls_get_correspondences.restype = C.c_int
        L.fls_get_correspondences.argtypes = [hp, C.c_int, ip, bp, bp, C.c_size_t]
        L.fls_map_size.restype = C.c_size_t
        L.fls_map_size.argtypes = [hp, C.c_int]
        L.fls_set_profiling.restype = C.c_int
        L.fls_set_profiling.argtypes = [hp, C.c_int]
        L.fls_get_kernel_time.restype = C.c_int
        L.fls_get_kernel_time.argtypes = [hp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.fls_get_traffic_counters.restype = C.c_int
        L.fls_get_traffic_counters.argtypes = [hp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.fls_map_export.restype = C.c_size_t
        L.fls_map_export.argtypes = [hp, C.c_void_p, C.c_size_t]
        L.fls_map_import.restype = C.c_int
        L.fls_map_import.argtypes = [hp, C.c_void_p, C.c_size_t]
        L.fls_loop_match.restype = C.c_int
        L.fls_loop_match.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                     C.c_void_p]
        L.fls_voxel_grid_cloud.restype = C.c_int
        L.fls_voxel_grid_cloud.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.fls_debug_voxel_grid.restype = C.c_int
        L.fls_debug_voxel_grid.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.fls_debug_voxel_grid_timed.restype = C.c_int
        L.fls_debug_voxel_grid_timed.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
        L.fls_debug_exact_sort.restype = C.c_int
        L.fls_debug_exact_sort.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.c_int]
        L.fls_debug_fullpiv_qr6.restype = C.c_int
        L.fls_debug_fullpiv_qr6.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
        L.fls_debug_ldlt6.restype = C.c_int
        L.fls_debug_ldlt6.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.fls_get_debug_stamps.restype = C.c_int
        L.fls_get_debug_stamps.argtypes = [hp, C.POINTER(C.c_int64)]
        L.fls_features_create.restype = C.c_int
        L.fls_features_create.argtypes = [C.POINTER(FeatureParams), C.c_int, C.POINTER(hp)]
        L.fls_features_destroy.restype = None
        L.fls_features_destroy.argtypes = [hp]
        L.fls_features_project.restype = C.c_int
        L.fls_features_project.argtypes = [hp, C.c_void_p, C.c_size_t, C.POINTER(PointLayout), C.POINTER(C.c_size_t)]
        L.fls_features_extract.restype = C.c_int
        L.fls_features_extract.argtypes = [hp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.fls_features_get.restype = C.c_size_t
        L.fls_features_get.argtypes = [hp, C.c_int, C.c_void_p, C.c_size_t]
        L.fls_features_get_time.restype = C.c_int
        L.fls_features_get_time.argtypes = [hp, dp, dp]
        L.fls_status_string.restype = C.c_char_p
        L.fls_status_string.argtypes = [C.c_int]
        L.fls_abi_version.restype = C.c_int
        L.fls_abi_revision.restype = C.c_int
        L.fls_device_count.restype = C.c_int
        _lib = L
    return _lib


def status_string(s: int) -> str:
    return lib().fls_status_string(int(s)).decode()


def device_count() -> int:
    return int(lib().fls_device_count())
