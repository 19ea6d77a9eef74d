"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg -- never by the product package
``funny_lidar_slam_amd``.  The oracle restates the reference's CPU algorithm
(see oracle/flo_oracle.cpp for the file:line map); it is the checker, not the
thing measured or shipped.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ICP_OPTIMIZED, P2PLANE_IVOX, INCREMENTAL_NDT, LOAM_FULL, P2PLANE_KDTREE = range(5)


class Params(C.Structure):
    """Field-for-field mirror of flo_params (== fls_params of include/fls_reg.h)."""

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


class Counters(C.Structure):
    _fields_ = [
        ("point_iters", C.c_uint64),
        ("probes", C.c_uint64),
        ("hit_voxels", C.c_uint64),
        ("cand_points", C.c_uint64),
        ("tie_queries", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (g++ only, no reference sources)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class FeatParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("vertical_scan", C.c_int32), ("horizontal_scan", C.c_int32),
                ("horizontal_resolution", C.c_float), ("min_distance", C.c_float), ("max_distance", C.c_float),
                ("corner_thres", C.c_float), ("planar_thres", C.c_float)]


# flo_feat_get selectors -> (numpy dtype, columns)
FEAT_ARRAYS = {"ordered": (0, np.float32, 4), "depth": (1, np.float32, 1), "col": (2, np.int32, 1), "row_start": (3, np.int32, 1),
               "row_end": (4, np.int32, 1), "corner": (5, np.float32, 4), "planar": (6, np.float32, 4), "is_corner": (7, np.uint8, 1),
               "roughness": (8, np.float32, 1), "valid_pre": (9, np.uint8, 1), "valid_post": (10, np.uint8, 1),
               "corner_idx": (11, np.int32, 1), "planar_idx": (12, np.int32, 1), "raw_index": (13, np.int32, 1)}


class OracleFeatures:
    """CPU oracle of PointcloudProjector::Project + FeatureExtractor::ExtractFeatures (oracle/flo_features.h)."""

    def __init__(self, vertical_scan, horizontal_scan, horizontal_resolution, min_distance, max_distance, corner_thres, planar_thres):
        self.params = FeatParams(C.sizeof(FeatParams), vertical_scan, horizontal_scan, horizontal_resolution, min_distance, max_distance,
                                 corner_thres, planar_thres)
        self._h = lib().flo_feat_create(C.byref(self.params))
        assert self._h

    def Project(self, raw: np.ndarray) -> int:
        """raw: structured array with fields x, y, z, intensity, ring (any itemsize / offsets)."""
        raw = np.ascontiguousarray(raw)
        f = raw.dtype.fields
        assert f["y"][1] == f["x"][1] + 4 and f["z"][1] == f["x"][1] + 8
        self._raw = raw
        return int(lib().flo_feat_project(self._h, raw.ctypes.data, raw.shape[0], raw.dtype.itemsize, f["x"][1], f["intensity"][1], f["ring"][1]))

    def ExtractFeatures(self) -> bool:
        return bool(lib().flo_feat_extract(self._h))

    def get(self, name: str) -> np.ndarray:
        what, dt, cols = FEAT_ARRAYS[name]
        n = lib().flo_feat_get(self._h, what, None, 0)
        out = np.zeros((max(n, 1), cols), dtype=dt)
        lib().flo_feat_get(self._h, what, out.ctypes.data, n)
        out = out[:n]
        return out if cols > 1 else out.reshape(-1)

    def tie_pairs(self) -> int:
        return int(lib().flo_feat_tie_pairs(self._h))

    def set_sort_mode(self, mode: int) -> None:
        """0: ties keep index order (default, what the HIP kernel reproduces); 1: libstdc++ std::sort order (= the reference's)."""
        lib().flo_feat_set_sort_mode(self._h, int(mode))

    def close(self):
        if self._h:
            lib().flo_feat_destroy(self._h)
            self._h = None


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        bp = C.POINTER(C.c_uint8)
        L.flo_create.restype = C.c_void_p
        L.flo_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.flo_destroy.argtypes = [C.c_void_p]
        L.flo_set_threads.argtypes = [C.c_int]
        L.flo_get_threads.restype = C.c_int
        L.flo_add_cloud.argtypes = [C.c_void_p, fp, C.c_size_t, fp, C.c_size_t, C.c_int]
        L.flo_match.argtypes = [C.c_void_p, fp, C.c_size_t, fp, C.c_size_t, C.c_int, dp, C.c_int, C.POINTER(Stats)]
        L.flo_fitness.restype = C.c_float
        L.flo_fitness.argtypes = [C.c_void_p, C.c_float]
        L.flo_get_iteration_log.argtypes = [C.c_void_p, dp, ip, dp, C.c_int]
        L.flo_get_correspondences.argtypes = [C.c_void_p, C.c_int, ip, bp, bp, C.c_size_t]
        L.flo_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.flo_get_tie_flags.restype = C.c_size_t
        L.flo_get_tie_flags.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t]
        L.flo_set_instrumentation.restype = None
        L.flo_set_instrumentation.argtypes = [C.c_void_p, C.c_int]
        L.flo_set_tie_break_by_id.restype = None
        L.flo_set_tie_break_by_id.argtypes = [C.c_int]
        L.flo_get_last_system.argtypes = [C.c_void_p, dp, dp]
        L.flo_map_size.restype = C.c_size_t
        L.flo_map_size.argtypes = [C.c_void_p, C.c_int]
        L.flo_map_voxels.restype = C.c_size_t
        L.flo_map_voxels.argtypes = [C.c_void_p]
        L.flo_set_ivox_capacity.restype = None
        L.flo_set_ivox_capacity.argtypes = [C.c_void_p, C.c_size_t]
        L.flo_map_dump.restype = C.c_size_t
        L.flo_map_dump.argtypes = [C.c_void_p, C.c_int, fp, C.c_size_t]
        L.flo_ndt_dump.restype = C.c_size_t
        L.flo_ndt_dump.argtypes = [C.c_void_p, ip, dp, dp, bp, ip, C.c_size_t]
        L.flo_loop_match.restype = C.c_float
        L.flo_loop_match.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, dp, C.POINTER(LoopStats)]
        L.flo_ndt_derivatives.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, C.c_float, dp, dp, dp, dp]
        L.flo_ndt_leaves.restype = C.c_size_t
        L.flo_ndt_leaves.argtypes = [fp, C.c_size_t, C.c_int, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, fp, C.c_size_t]
        L.flo_gicp_covariances.restype = None
        L.flo_gicp_covariances.argtypes = [fp, C.c_size_t, C.c_int, C.c_int, C.c_double, dp]
        L.flo_gicp_fdf.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, dp, C.c_double, dp, dp, dp, C.POINTER(C.c_int32)]
        L.flo_jacobi_svd_solve6.restype = None
        L.flo_jacobi_svd_solve6.argtypes = [dp, dp, dp]
        L.flo_voxel_grid.restype = C.c_size_t
        L.flo_voxel_grid.argtypes = [fp, C.c_size_t, C.c_int, C.c_float, fp]
        L.flo_so3_exp.argtypes = [dp, dp]
        L.flo_so3_hat.argtypes = [dp, dp]
        L.flo_rpy.argtypes = [dp, dp]
        L.flo_colpiv_qr_solve_5x3.argtypes = [dp, dp, dp]
        L.flo_fullpiv_qr_solve_6.argtypes = [dp, dp, dp]
        L.flo_lu_inverse_6.argtypes = [dp, dp, dp]
        L.flo_inverse3.argtypes = [dp, dp]
        L.flo_svd3.argtypes = [dp, dp, dp, dp]
        L.flo_knn_bruteforce.argtypes = [fp, C.c_size_t, fp, C.c_int, ip, fp]
        L.flo_kdtree_knn.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, ip, fp]
        L.flo_feat_create.restype = C.c_void_p
        L.flo_feat_create.argtypes = [C.POINTER(FeatParams)]
        L.flo_feat_destroy.argtypes = [C.c_void_p]
        L.flo_feat_project.restype = C.c_int64
        L.flo_feat_project.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        L.flo_feat_extract.argtypes = [C.c_void_p]
        L.flo_feat_get.restype = C.c_size_t
        L.flo_feat_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.flo_feat_tie_pairs.restype = C.c_uint64
        L.flo_feat_tie_pairs.argtypes = [C.c_void_p]
        L.flo_feat_set_sort_mode.restype = None
        L.flo_feat_set_sort_mode.argtypes = [C.c_void_p, C.c_int]
        L.flo_col_index.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        L.flo_fast_atan2f.restype = C.c_float
        L.flo_fast_atan2f.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _cloud(c):
    """(n,3) or (n,4) float32 -> (array, ptr, n, stride)."""
    if c is None:
        return None, None, 0, 0
    a = np.ascontiguousarray(c, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] in (3, 4, 8)
    return a, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]


class OracleMatcher:
    """Python mirror of RegistrationInterface backed by the CPU oracle."""

    K = {ICP_OPTIMIZED: 1, P2PLANE_IVOX: 5, INCREMENTAL_NDT: 7, LOAM_FULL: 5, P2PLANE_KDTREE: 5}

    def __init__(self, kind: int, params: Params):
        self.kind = kind
        self.params = params
        self._h = lib().flo_create(kind, C.byref(params))
        if not self._h:
            raise RuntimeError("flo_create failed")
        self.stats = Stats()
        self._n = (0, 0)

    def close(self):
        if self._h:
            lib().flo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AddCloudToLocalMap(self, cloud0, cloud1=None):
        a0, p0, n0, s0 = _cloud(cloud0)
        a1, p1, n1, s1 = _cloud(cloud1)
        if a1 is not None:
            assert s1 == s0
        return lib().flo_add_cloud(self._h, p0, n0, p1, n1, s0)

    def Match(self, src0, T, src1=None, update_map=True):
        """T: (4,4) float64 world<-body, numpy row-major view; returns (ok, T_out)."""
        a0, p0, n0, s0 = _cloud(src0)
        a1, p1, n1, s1 = _cloud(src1)
        Tc = np.asfortranarray(np.asarray(T, dtype=np.float64)).copy(order="F")
        flat = np.ascontiguousarray(Tc.reshape(-1, order="F"))
        rc = lib().flo_match(self._h, p0, n0, p1, n1, s0, flat.ctypes.data_as(C.POINTER(C.c_double)),
                             1 if update_map else 0, C.byref(self.stats))
        self._n = (n0, n1)
        return rc == 0, flat.reshape(4, 4, order="F").copy()

    def GetFitnessScore(self, max_range: float) -> float:
        return float(lib().flo_fitness(self._h, max_range))

    def iteration_log(self, cap=64):
        T = np.zeros((cap, 16)); nv = np.zeros(cap, np.int32); sr = np.zeros(cap)
        n = lib().flo_get_iteration_log(self._h, T.ctypes.data_as(C.POINTER(C.c_double)),
                                        nv.ctypes.data_as(C.POINTER(C.c_int32)),
                                        sr.ctypes.data_as(C.POINTER(C.c_double)), cap)
        n = min(n, cap)
        Ts = np.stack([T[i].reshape(4, 4, order="F") for i in range(n)]) if n else np.zeros((0, 4, 4))
        return Ts, nv[:n].copy(), sr[:n].copy()

    def correspondences(self, slot=0):
        n = self.stats.n_source_corner if slot == 1 else self.stats.n_source
        k = self.K[self.kind]
        ids = np.full((n, k), -1, np.int32); cnt = np.zeros(n, np.uint8); valid = np.zeros(n, np.uint8)
        if n:
            lib().flo_get_correspondences(self._h, slot, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                          cnt.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          valid.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return ids, cnt, valid

    def set_instrumentation(self, on: bool) -> None:
        """False: no traffic / tie counters in the kNN stage (timing runs: cpu_baseline)."""
        lib().flo_set_instrumentation(self._h, 1 if on else 0)

    def tie_rows(self):
        """bool per query of the last Match: the row's neighbour list was decided by an exact distance tie (None: no tie flags for this kind)."""
        n = lib().flo_get_tie_flags(self._h, None, 0)
        if n == 0:
            return None
        out = np.zeros(n, np.uint8)
        lib().flo_get_tie_flags(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return out.astype(bool)

    def counters(self) -> Counters:
        c = Counters()
        lib().flo_get_counters(self._h, C.byref(c))
        return c

    def last_system(self):
        H = np.zeros(36); g = np.zeros(6)
        lib().flo_get_last_system(self._h, H.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)))
        return H.reshape(6, 6, order="F"), g

    def map_size(self, slot=0):
        return int(lib().flo_map_size(self._h, slot))

    def map_voxels(self):
        return int(lib().flo_map_voxels(self._h))

    def set_ivox_capacity(self, cap: int):
        lib().flo_set_ivox_capacity(self._h, int(cap))

    def map_dump(self, slot=0):
        n = self.map_size(slot)
        out = np.zeros((max(n, 1), 3), np.float32)
        m = lib().flo_map_dump(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out[: min(n, m)]

    def ndt_dump(self):
        n = self.map_size(0)
        keys = np.zeros((n, 3), np.int32); mu = np.zeros((n, 3)); info = np.zeros((n, 9))
        est = np.zeros(n, np.uint8); npts = np.zeros(n, np.int32)
        lib().flo_ndt_dump(self._h, keys.ctypes.data_as(C.POINTER(C.c_int32)), mu.ctypes.data_as(C.POINTER(C.c_double)),
                           info.ctypes.data_as(C.POINTER(C.c_double)), est.ctypes.data_as(C.POINTER(C.c_uint8)),
                           npts.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return keys, mu, info.reshape(n, 3, 3).transpose(0, 2, 1).copy(), est, npts


def set_threads(n: int):
    lib().flo_set_threads(int(n))


def get_threads() -> int:
    return int(lib().flo_get_threads())


# ---- stand-alone helpers -------------------------------------------------------------------
def so3_exp(v):
    a, p = _f64(v); R = np.zeros(9)
    lib().flo_so3_exp(p, R.ctypes.data_as(C.POINTER(C.c_double)))
    return R.reshape(3, 3, order="F")


def so3_hat(v):
    a, p = _f64(v); R = np.zeros(9)
    lib().flo_so3_hat(p, R.ctypes.data_as(C.POINTER(C.c_double)))
    return R.reshape(3, 3, order="F")


def rpy(R):
    a, p = _f64(np.asarray(R, dtype=np.float64).reshape(-1, order="F")); o = np.zeros(3)
    lib().flo_rpy(p, o.ctypes.data_as(C.POINTER(C.c_double)))
    return o


def colpiv_qr_solve_5x3(A, b):
    a, pa = _f64(np.asarray(A, dtype=np.float64).reshape(-1, order="F")); bb, pb = _f64(b); x = np.zeros(3)
    lib().flo_colpiv_qr_solve_5x3(pa, pb, x.ctypes.data_as(C.POINTER(C.c_double)))
    return x


def fullpiv_qr_solve_6(A, b):
    a, pa = _f64(np.asarray(A, dtype=np.float64).reshape(-1, order="F")); bb, pb = _f64(b); x = np.zeros(6)
    lib().flo_fullpiv_qr_solve_6(pa, pb, x.ctypes.data_as(C.POINTER(C.c_double)))
    return x


def lu_inverse_6(A):
    a, pa = _f64(np.asarray(A, dtype=np.float64).reshape(-1, order="F")); inv = np.zeros(36); det = C.c_double()
    lib().flo_lu_inverse_6(pa, inv.ctypes.data_as(C.POINTER(C.c_double)), C.byref(det))
    return inv.reshape(6, 6, order="F"), det.value


def inverse3(A):
    a, pa = _f64(np.asarray(A, dtype=np.float64).reshape(-1, order="F")); inv = np.zeros(9)
    lib().flo_inverse3(pa, inv.ctypes.data_as(C.POINTER(C.c_double)))
    return inv.reshape(3, 3, order="F")


def svd3(A):
    a, pa = _f64(np.asarray(A, dtype=np.float64).reshape(-1, order="F"))
    U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
    lib().flo_svd3(pa, U.ctypes.data_as(C.POINTER(C.c_double)), S.ctypes.data_as(C.POINTER(C.c_double)),
                   V.ctypes.data_as(C.POINTER(C.c_double)))
    return U.reshape(3, 3, order="F"), S, V.reshape(3, 3, order="F")


class LoopStats(C.Structure):
    """flo_loop_stats (flo_api.h) == fls_loop_stats (include/fls_reg.h)"""
    _fields_ = [("ndt_iterations", C.c_int32 * 4), ("ndt_evaluations", C.c_int32 * 4), ("ndt_source_points", C.c_int32 * 4), ("ndt_target_leaves", C.c_int32 * 4),
                ("gicp_iterations", C.c_int32), ("gicp_inner_iterations", C.c_int32), ("gicp_evaluations", C.c_int32), ("gicp_correspondences", C.c_int32),
                ("gicp_source_points", C.c_int32), ("gicp_target_points", C.c_int32), ("gicp_failed", C.c_int32), ("reserved", C.c_int32),
                ("ndt_score", C.c_double * 4), ("T_after_ndt", C.c_double * 16)]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def loop_match(source, target, T_init):
    """LoopClosure::Match (loop_closure.cpp:233-267): returns (fitness, T (4,4), LoopStats)"""
    a, pa, na, sa = _cloud(source)
    b, pb, nb, sb = _cloud(target)
    assert sa == sb
    T = np.ascontiguousarray(np.asarray(T_init, np.float64).reshape(4, 4).T).reshape(-1).copy()
    st = LoopStats()
    f = lib().flo_loop_match(pa, na, pb, nb, sa, _dp(T), C.byref(st))
    return float(f), T.reshape(4, 4).T.copy(), st


def ndt_derivatives(source, target, resolution, p):
    a, pa, na, sa = _cloud(source)
    b, pb, nb, sb = _cloud(target)
    p = np.ascontiguousarray(p, np.float64)
    score = C.c_double()
    g = np.zeros(6); H = np.zeros(36)
    rc = lib().flo_ndt_derivatives(pa, na, pb, nb, sa, resolution, _dp(p), C.byref(score), _dp(g), _dp(H))
    if rc != 0:
        raise RuntimeError("flo_ndt_derivatives: no leaf with >= 6 points")
    return score.value, g, H.reshape(6, 6, order="F")


def ndt_leaves(target, resolution):
    b, pb, nb, sb = _cloud(target)
    cap = max(nb, 1)
    idx = np.zeros(cap, np.int32); nr = np.zeros(cap, np.int32); mean = np.zeros((cap, 3)); icov = np.zeros((cap, 9)); cen = np.zeros((cap, 3), np.float32)
    n = lib().flo_ndt_leaves(pb, nb, sb, resolution, idx.ctypes.data_as(C.POINTER(C.c_int32)), nr.ctypes.data_as(C.POINTER(C.c_int32)), _dp(mean), _dp(icov),
                             cen.ctypes.data_as(C.POINTER(C.c_float)), cap)
    return idx[:n], nr[:n], mean[:n], icov[:n].reshape(n, 3, 3).transpose(0, 2, 1), cen[:n]


def gicp_covariances(cloud, k=20, eps=0.001):
    a, pa, na, sa = _cloud(cloud)
    out = np.zeros((na, 9))
    lib().flo_gicp_covariances(pa, na, sa, k, eps, _dp(out))
    return out.reshape(na, 3, 3).transpose(0, 2, 1)


def gicp_fdf(source, target, guess, corr_dist, x):
    a, pa, na, sa = _cloud(source)
    b, pb, nb, sb = _cloud(target)
    G = np.ascontiguousarray(np.asarray(guess, np.float64).reshape(4, 4).T).reshape(-1).copy()
    x = np.ascontiguousarray(x, np.float64)
    f = C.c_double(); g = np.zeros(6); nc = C.c_int32()
    rc = lib().flo_gicp_fdf(pa, na, pb, nb, sa, _dp(G), corr_dist, _dp(x), C.byref(f), _dp(g), C.byref(nc))
    if rc != 0:
        raise RuntimeError("flo_gicp_fdf: too few points / correspondences")
    return f.value, g, nc.value


def jacobi_svd_solve6(A, b):
    A = np.ascontiguousarray(np.asarray(A, np.float64).T).reshape(-1).copy()
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(6)
    lib().flo_jacobi_svd_solve6(_dp(A), _dp(b), _dp(x))
    return x


def voxel_grid(cloud, leaf):
    a, p, n, s = _cloud(cloud)
    out = np.zeros((max(n, 1), 4), np.float32)
    m = lib().flo_voxel_grid(p, n, s, leaf, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:m].copy()


def kdtree_knn(map_xyz, queries, k):
    m, pm = _f32(map_xyz); q, pq = _f32(queries)
    idx = np.zeros((q.shape[0], k), np.int32); d2 = np.zeros((q.shape[0], k), np.float32)
    lib().flo_kdtree_knn(pm, m.shape[0], pq, q.shape[0], k, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                         d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d2


def knn_bruteforce(map_xyz, q, k):
    m, pm = _f32(map_xyz); qq, pq = _f32(q)
    idx = np.zeros(k, np.int32); d2 = np.zeros(k, np.float32)
    n = lib().flo_knn_bruteforce(pm, m.shape[0], pq, k, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                 d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx[:n], d2[:n]


def set_tie_break_by_id(on: bool) -> None:
    """TEST SWITCH (process-wide): exact distance ties of the iVox kNN ordered by insertion id like the device's keys, not by introselect (flo_api.h)."""
    lib().flo_set_tie_break_by_id(1 if on else 0)

