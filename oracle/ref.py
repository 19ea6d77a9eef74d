"""ctypes front-end of oracle/_ref/libref.so: the REFERENCE'S OWN registration classes, compiled verbatim from
/root/reference by oracle/ref_shim/Makefile against an include-shadow shim (Eigen / PCL / glog stand-ins).

TEST INFRASTRUCTURE ONLY -- it exists to PIN the CPU oracle (tests/test_ref_pin.py); nothing in the product, in the
`-m gpu` tests, in smoke() or in bench.py needs it, and it can only be (re)built where /root/reference exists.

The reference keeps function-static state (``is_first``, ``last_T``: SURVEY Q12), so a process may hold ONE matcher of
a kind, used for one scenario: tests/refpin.py drives this module from a fresh worker process per scenario.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import (FEAT_ARRAYS, FeatParams, ICP_OPTIMIZED, INCREMENTAL_NDT, LOAM_FULL, P2PLANE_IVOX, P2PLANE_KDTREE, Params, Stats,  # noqa: F401
                     _cloud)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLS_REF_PAR=1: the build whose parallel-STL loops (std::execution::par / par_unseq for_each) run on OpenMP threads
# (oracle/ref_shim/include/pstl_omp.hpp stands in for the TBB backend the reference links); same sources, same results
REFERENCE_ROOT = "/root/reference"
PARALLEL = os.environ.get("FLS_REF_PAR", "0") == "1" and (os.path.exists(os.path.join(_HERE, "_ref", "libref_par.so")) or
                                                          os.path.isdir(os.path.join(REFERENCE_ROOT, "include", "registration")))
_LIB_PATH = os.path.join(_HERE, "_ref", "libref_par.so" if PARALLEL else "libref.so")
_lib = None


def available() -> bool:
    """libref.so is present (built here, or shipped with the snapshot), or can be built (reference checkout present)."""
    return os.path.exists(_LIB_PATH) or os.path.isdir(os.path.join(REFERENCE_ROOT, "include", "registration"))


def build(force: bool = False) -> str:
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "include", "registration")):
        # only the two libraries: the Makefile's `all` also links tests/harness/gpu_vs_ref against libfls_reg.so, which a tree that has built
        # nothing but the oracle does not have (the first pin test died in `make all` there)
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_shim"), "-s", "../_ref/libref.so", "../_ref/libref_par.so"] + (["-B"] if force else []))
    if not os.path.exists(_LIB_PATH):
        raise FileNotFoundError(f"{_LIB_PATH}: not built and {REFERENCE_ROOT} is not present")
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, ip, bp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_last_error.restype = C.c_char_p
        L.ref_last_error.argtypes = [C.c_void_p]
        L.ref_set_ivox_capacity.restype = None
        L.ref_set_ivox_capacity.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_add_cloud.argtypes = [C.c_void_p, fp, C.c_size_t, fp, C.c_size_t, C.c_int]
        L.ref_match.argtypes = [C.c_void_p, fp, C.c_size_t, fp, C.c_size_t, C.c_int, dp, C.POINTER(Stats)]
        L.ref_fitness.restype = C.c_float
        L.ref_fitness.argtypes = [C.c_void_p, C.c_float]
        L.ref_map_size.restype = C.c_size_t
        L.ref_map_size.argtypes = [C.c_void_p, C.c_int]
        L.ref_map_voxels.restype = C.c_size_t
        L.ref_map_voxels.argtypes = [C.c_void_p]
        L.ref_map_dump.restype = C.c_size_t
        L.ref_map_dump.argtypes = [C.c_void_p, C.c_int, fp, ip, C.c_size_t]
        L.ref_get_flags.argtypes = [C.c_void_p, C.c_int, bp, dp, C.c_size_t]
        L.ref_get_nearest.argtypes = [C.c_void_p, fp, bp, C.c_size_t]
        L.ref_last_system.argtypes = [C.c_void_p, dp, dp]
        L.ref_ndt_dump.restype = C.c_size_t
        L.ref_ndt_dump.argtypes = [C.c_void_p, ip, dp, dp, dp, bp, ip, ip, C.c_size_t]
        L.ref_feat_create.restype = C.c_void_p
        L.ref_feat_create.argtypes = [C.POINTER(FeatParams)]
        L.ref_feat_destroy.argtypes = [C.c_void_p]
        L.ref_feat_project.restype = C.c_int64
        L.ref_feat_project.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        L.ref_feat_extract.argtypes = [C.c_void_p]
        L.ref_feat_get.restype = C.c_size_t
        L.ref_feat_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_col_index.argtypes = [C.c_float, C.c_float, C.c_char_p]
        L.ref_fast_atan2f.restype = C.c_float
        L.ref_fast_atan2f.argtypes = [C.c_float, C.c_float]
        L.ref_set_threads.restype = C.c_int
        L.ref_set_threads.argtypes = [C.c_int]
        L.ref_so3_exp.argtypes = [dp, dp]
        L.ref_rpy.argtypes = [dp, dp]
        _lib = L
    return _lib


class RefMatcher:
    """The reference's RegistrationInterface implementation of `kind` (one per kind per process, see module doc)."""

    def __init__(self, kind: int, params: Params):
        self.kind = kind
        self.params = params
        self._h = lib().ref_create(kind, C.byref(params))
        if not self._h:
            raise RuntimeError("ref_create failed (a CHECK in the reference's constructor fired)")
        self.stats = Stats()

    def close(self):
        if self._h:
            lib().ref_destroy(self._h)
            self._h = None

    def _err(self, what):
        return RuntimeError(f"{what}: {lib().ref_last_error(self._h).decode(errors='replace')}")

    def set_ivox_capacity(self, cap: int):
        lib().ref_set_ivox_capacity(self._h, int(cap))

    def AddCloudToLocalMap(self, cloud0, cloud1=None):
        a0, p0, n0, s0 = _cloud(cloud0)
        a1, p1, n1, s1 = _cloud(cloud1)
        if lib().ref_add_cloud(self._h, p0, n0, p1, n1, s0) != 0:
            raise self._err("AddCloudToLocalMap")

    def Match(self, src0, T, src1=None):
        a0, p0, n0, s0 = _cloud(src0)
        a1, p1, n1, s1 = _cloud(src1)
        flat = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).reshape(-1, order="F"))
        rc = lib().ref_match(self._h, p0, n0, p1, n1, s0, flat.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.stats))
        if rc < 0:
            raise self._err("Match")
        return rc == 0, flat.reshape(4, 4, order="F").copy()

    def GetFitnessScore(self, max_range: float) -> float:
        return float(lib().ref_fitness(self._h, max_range))

    def map_size(self, slot=0) -> int:
        return int(lib().ref_map_size(self._h, slot))

    def map_voxels(self) -> int:
        return int(lib().ref_map_voxels(self._h))

    def map_dump(self, slot=0):
        n = self.map_size(slot)
        xyzi = np.zeros((max(n, 1), 4), np.float32); keys = np.zeros((max(n, 1), 3), np.int32)
        lib().ref_map_dump(self._h, slot, xyzi.ctypes.data_as(C.POINTER(C.c_float)), keys.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return xyzi[:n], keys[:n]

    def flags(self, slot=0):
        n = self.stats.n_source_corner if slot == 1 else self.stats.n_source
        valid = np.zeros(max(n, 1), np.uint8); res = np.zeros(max(n, 1))
        m = lib().ref_get_flags(self._h, slot, valid.ctypes.data_as(C.POINTER(C.c_uint8)), res.ctypes.data_as(C.POINTER(C.c_double)), n)
        assert m == n or m < 0, (m, n)
        return valid[:n], res[:n]

    def nearest(self):
        n = self.stats.n_source
        xyz = np.zeros((max(n, 1), 5, 3), np.float32); cnt = np.zeros(max(n, 1), np.uint8)
        m = lib().ref_get_nearest(self._h, xyz.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return xyz[:n], cnt[:n], m

    def last_system(self):
        H = np.zeros(36); g = np.zeros(6)
        lib().ref_last_system(self._h, H.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)))
        return H.reshape(6, 6, order="F"), g

    def ndt_dump(self):
        n = self.map_size(0)
        keys = np.zeros((n, 3), np.int32); mu = np.zeros((n, 3)); sigma = np.zeros((n, 9)); info = np.zeros((n, 9))
        est = np.zeros(n, np.uint8); npts = np.zeros(n, np.int32); pend = np.zeros(n, np.int32)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        lib().ref_ndt_dump(self._h, keys.ctypes.data_as(ip), mu.ctypes.data_as(dp), sigma.ctypes.data_as(dp), info.ctypes.data_as(dp),
                           est.ctypes.data_as(C.POINTER(C.c_uint8)), npts.ctypes.data_as(ip), pend.ctypes.data_as(ip), n)
        tr = lambda a: a.reshape(n, 3, 3).transpose(0, 2, 1).copy()  # noqa: E731  (col-major -> row-major)
        return dict(keys=keys, mu=mu, sigma=tr(sigma), info=tr(info), est=est, npts=npts, pending=pend)


class RefFeatures:
    """The reference's loam::PointcloudProjector + loam::FeatureExtractor (de-skew = identity, see ref_shim)."""

    NAMES = ("ordered", "depth", "col", "row_start", "row_end", "corner", "planar", "is_corner", "valid_post")  # what the reference keeps

    def __init__(self, vertical_scan, horizontal_scan, horizontal_resolution, min_distance, max_distance, corner_thres, planar_thres):
        self.p = FeatParams(C.sizeof(FeatParams), vertical_scan, horizontal_scan, horizontal_resolution, min_distance, max_distance, corner_thres,
                            planar_thres)
        self._h = lib().ref_feat_create(C.byref(self.p))
        if not self._h:
            raise RuntimeError("ref_feat_create failed")

    def Project(self, raw: np.ndarray) -> int:
        raw = np.ascontiguousarray(raw)
        f = raw.dtype.fields
        n = lib().ref_feat_project(self._h, raw.ctypes.data, raw.shape[0], raw.dtype.itemsize, f["x"][1], f["intensity"][1], f["ring"][1], f["time"][1])
        if n < 0:
            raise RuntimeError("Project failed")
        return int(n)

    def ExtractFeatures(self) -> bool:
        return lib().ref_feat_extract(self._h) == 1

    def get(self, name: str) -> np.ndarray:
        what, dt, cols = FEAT_ARRAYS[name]
        n = lib().ref_feat_get(self._h, what, None, 0)
        out = np.zeros((max(n, 1), cols), dtype=dt)
        lib().ref_feat_get(self._h, what, out.ctypes.data, n)
        out = out[:n]
        return out if cols > 1 else out.reshape(-1)

    def close(self):
        if self._h:
            lib().ref_feat_destroy(self._h)
            self._h = None
