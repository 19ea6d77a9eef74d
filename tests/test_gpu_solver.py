"""The Gauss-Newton tail's 6x6 solver on the device (wave_solve.hpp::fullpiv_qr_solve6_wave, Eigen FullPivHouseholderQR::solve
semantics: loam_point_to_plane_ivox.h:167, loam_full_kdtree.h:141, loam_point_to_plane_kdtree.h:108) against the oracle's
restatement, BIT FOR BIT: well-conditioned normal equations, rank-deficient ones (all normals parallel, planar scenes), exact
ties of the pivot search (symmetric matrices tie H(i,j) with H(j,i) by construction), zero matrices, wild scales."""
import ctypes as C

import numpy as np
import pytest

from funny_lidar_slam_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _systems():
    rng = np.random.default_rng(20241022)
    Hs, gs = [], []

    def add(J, r):
        Hs.append(J.T @ J); gs.append(-J.T @ r)

    for k in range(3000):
        kind = k % 10
        n = int(rng.integers(6, 400))
        J = rng.normal(size=(n, 6)) * rng.choice([1e-3, 1.0, 50.0], size=6)
        if kind == 1: J[:, 3:] = np.outer(rng.normal(size=n), [0.0, 0.0, 1.0])                 # all normals parallel (a single plane): rank 3-4
        if kind == 2: J[:, 5] = 0.0                                                             # an unobserved direction
        if kind == 3: J[:, 4] = J[:, 3] * 2.0                                                   # exactly dependent columns
        if kind == 4: J = J[:2]                                                                 # two residuals only
        if kind == 5: J *= 0.0                                                                  # nothing valid: H = 0
        if kind == 6: J[:, 0] *= 1e-9                                                           # a nearly negligible pivot
        if kind == 7: J = np.round(J)                                                           # small integers: exact ties in the pivot search
        r = rng.normal(size=J.shape[0]) * 0.05
        add(J, r)
    H = np.stack(Hs); g = np.stack(gs)
    H = 0.5 * (H + H.transpose(0, 2, 1))  # exactly symmetric, like the device's upper-triangle assembly
    return H, g


def test_fullpiv_qr6_bit_exact_against_oracle(built):
    assert _lib.device_count() >= 1
    H, g = _systems()
    n = H.shape[0]
    Hc = np.ascontiguousarray(H.transpose(0, 2, 1)).reshape(n, 36)  # column-major per system
    x = np.zeros((n, 6))
    dp = C.POINTER(C.c_double)
    rc = _lib.lib().fls_debug_fullpiv_qr6(0, Hc.ctypes.data_as(dp), np.ascontiguousarray(g).ctypes.data_as(dp), n, x.ctypes.data_as(dp))
    assert rc == 0
    bad = 0
    for s in range(n):
        ref = O.fullpiv_qr_solve_6(H[s], g[s])
        if not (np.array_equal(ref, x[s]) or (np.isnan(ref).any() and np.isnan(x[s]).any())):
            bad += 1
            if bad < 5:
                print(s, s % 10, ref, x[s])
    assert bad == 0, bad
    assert int((np.abs(x).sum(1) == 0).sum()) >= 250  # the zero systems (and only exact zeros there)
