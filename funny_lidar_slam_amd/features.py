"""LOAM feature front-end on the GPU: Python mirror of the reference's two classes (same constructor arguments,
same PointcloudCluster fields), bound to include/fls_features.h through ctypes.

    loam::PointcloudProjector   include/loam/pointcloud_projector.h:15-38, src/loam/pointcloud_projector.cpp:32-133
    loam::FeatureExtractor      include/loam/feature_extractor.h:15-45,    src/loam/feature_extractor.cpp:36-222

The projection stays resident on the device between `Project` and `ExtractFeatures` (the cluster carries the handle).
De-skew is outside this library (pass corrected points).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FeatureParams, FlsError, PointLayout

ARRAYS = {"ordered": (0, np.float32, 4), "depth": (1, np.float32, 1), "col": (2, np.int32, 1), "row_start": (3, np.int32, 1),
          "row_end": (4, np.int32, 1), "corner": (5, np.float32, 4), "planar": (6, np.float32, 4), "is_corner": (7, np.uint8, 1),
          "roughness": (8, np.float32, 1), "valid_pre": (9, np.uint8, 1), "valid_post": (10, np.uint8, 1), "corner_idx": (11, np.int32, 1),
          "planar_idx": (12, np.int32, 1), "raw_index": (13, np.int32, 1), "corner_filtered": (14, np.float32, 4),
          "planar_filtered": (15, np.float32, 4)}


class FeatureFrontEnd:
    """One fls_features handle (projector + extractor state on one device)."""

    def __init__(self, lidar_horizontal_scan, lidar_vertical_scan, lidar_horizontal_resolution, min_distance, max_distance,
                 corner_thres, planar_thres, corner_voxel_filter_size=0.0, planar_voxel_filter_size=0.0, device_id=0):
        self.params = FeatureParams(C.sizeof(FeatureParams), lidar_vertical_scan, lidar_horizontal_scan, lidar_horizontal_resolution,
                                    min_distance, max_distance, corner_thres, planar_thres, corner_voxel_filter_size, planar_voxel_filter_size)
        self._h = C.c_void_p()
        rc = _lib.lib().fls_features_create(C.byref(self.params), device_id, C.byref(self._h))
        if rc != _lib.FLS_OK:
            self._h = C.c_void_p()
            raise FlsError(rc, "fls_features_create")

    def project(self, raw: np.ndarray) -> int:
        raw = np.ascontiguousarray(raw)
        f = raw.dtype.fields
        if f is None or not all(k in f for k in ("x", "y", "z", "intensity", "ring")):
            raise ValueError("raw cloud must be a structured array with x, y, z, intensity, ring")
        if f["y"][1] != f["x"][1] + 4 or f["z"][1] != f["x"][1] + 8 or raw.dtype["ring"] != np.uint16:
            raise ValueError("x, y, z must be consecutive floats and ring a uint16")
        lay = PointLayout(raw.dtype.itemsize, f["x"][1], f["intensity"][1], f["ring"][1])
        n = C.c_size_t()
        rc = _lib.lib().fls_features_project(self._h, raw.ctypes.data, raw.shape[0], C.byref(lay), C.byref(n))
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_features_project")
        return int(n.value)

    def extract(self):
        nc, npl = C.c_size_t(), C.c_size_t()
        rc = _lib.lib().fls_features_extract(self._h, C.byref(nc), C.byref(npl))
        if rc != _lib.FLS_OK:
            raise FlsError(rc, "fls_features_extract")
        return int(nc.value), int(npl.value)

    def get(self, name: str) -> np.ndarray:
        what, dt, cols = ARRAYS[name]
        n = _lib.lib().fls_features_get(self._h, what, None, 0)
        out = np.zeros((max(n, 1), cols), dtype=dt)
        _lib.lib().fls_features_get(self._h, what, out.ctypes.data, n)
        out = out[:n]
        return out if cols > 1 else out.reshape(-1)

    def times_ms(self):
        a, b = C.c_double(), C.c_double()
        _lib.lib().fls_features_get_time(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().fls_features_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PointcloudProjector:
    """loam::PointcloudProjector(corrector, lidar_horizontal_scan, lidar_vertical_scan, lidar_horizontal_resolution,
    min_distance, max_distance) -- pointcloud_projector.cpp:14-30 (the distortion corrector is the identity here).
    The feature thresholds are needed when the device handle is created, so they are keyword arguments."""

    def __init__(self, lidar_horizontal_scan, lidar_vertical_scan, lidar_horizontal_resolution, min_distance, max_distance, *,
                 corner_thres=1.0, planar_thres=0.1, corner_voxel_filter_size=0.0, planar_voxel_filter_size=0.0, device_id=0):
        self.front = FeatureFrontEnd(lidar_horizontal_scan, lidar_vertical_scan, lidar_horizontal_resolution, min_distance, max_distance,
                                     corner_thres, planar_thres, corner_voxel_filter_size, planar_voxel_filter_size, device_id)

    def Project(self, cluster) -> None:
        self.front.project(cluster.raw_cloud_)
        cluster.ordered_cloud_ = self.front.get("ordered")
        cluster.point_depth_vec_ = self.front.get("depth")
        cluster.point_col_index_vec_ = self.front.get("col")
        cluster.row_start_index_vec_ = self.front.get("row_start")
        cluster.row_end_index_vec_ = self.front.get("row_end")
        cluster.feature_state_ = self.front


class FeatureExtractor:
    """loam::FeatureExtractor(corner_thr, planar_thr, lidar_horizontal_scan, lidar_vertical_scan) -- feature_extractor.cpp:16-29.
    Continues from the projection the cluster carries (thresholds must equal the ones the handle was created with)."""

    def __init__(self, corner_thr, planar_thr, lidar_horizontal_scan, lidar_vertical_scan):
        self.corner_thr, self.planar_thr = float(np.float32(corner_thr)), float(np.float32(planar_thr))
        self.cols, self.rows = lidar_horizontal_scan, lidar_vertical_scan

    def ExtractFeatures(self, cluster) -> None:
        front = cluster.feature_state_
        if front is None:
            raise FlsError(_lib.FLS_ERR_STATE, "ExtractFeatures before Project")
        p = front.params
        if (p.corner_thres, p.planar_thres, p.lidar_horizontal_scan, p.lidar_vertical_scan) != (self.corner_thr, self.planar_thr, self.cols, self.rows):
            raise ValueError("FeatureExtractor parameters differ from the projector's device handle")
        front.extract()
        cluster.corner_cloud_ = front.get("corner")
        cluster.planar_cloud_ = front.get("planar")
