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


def _ldlt_pivots(H):
    """pivots of the unpivoted LDL^T of a symmetric 6x6 (the rule the fast path applies: hm::ldlt_solve6 / ldlt_solve6_wave)"""
    A = H.astype(np.float64).copy()
    d = np.zeros(6)
    with np.errstate(all="ignore"):
        for j in range(6):
            d[j] = A[j, j]
            l = A[j + 1:, j] / d[j]
            A[j + 1:, j + 1:] -= np.outer(l, A[j, j + 1:])
    return d


def test_ldlt6_fast_path_against_numpy(built):
    """The SPD fast path of the tails (wave_solve.hpp::ldlt_solve6_wave: rows in lanes, pivot rows by v_readlane, Newton reciprocals): it accepts
    exactly the systems the rule names (every pivot > 0, d_min > 1e-9 d_max: a numpy model of the factorisation, borderline ratios aside), never a
    rank-deficient or zero one (those belong to the restated Eigen solver, whose rank rule is the reference's semantics), and where it accepts,
    its solution is numpy's to the conditioning of the system."""
    assert _lib.device_count() >= 1
    H, g = _systems()
    n = H.shape[0]
    Hc = np.ascontiguousarray(H.transpose(0, 2, 1)).reshape(n, 36)
    x = np.zeros((n, 6)); ok = np.zeros(n, dtype=np.int32)
    dp = C.POINTER(C.c_double)
    rc = _lib.lib().fls_debug_ldlt6(0, Hc.ctypes.data_as(dp), np.ascontiguousarray(g).ctypes.data_as(dp), n, x.ctypes.data_as(dp), ok.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    kinds = np.arange(n) % 10
    for k in (1, 2, 3, 4, 5):  # rank-deficient by construction: never the fast path
        assert not ok[kinds == k].any(), k
    checked = wrong = 0
    for s in range(n):
        d = _ldlt_pivots(H[s])
        if not np.isfinite(d).all():
            assert not ok[s], s
            continue
        pos = bool((d > 0).all())
        ratio = d.min() / d.max() if pos else 0.0
        if pos and not (0.5e-9 < ratio < 2e-9):  # (a ratio at the threshold may fall either way under another rounding)
            checked += 1
            wrong += int(bool(ok[s]) != (ratio > 1e-9))
        elif not pos and d.min() < -1e-6 * np.abs(d).max():
            assert not ok[s], (s, d)
    assert wrong == 0 and checked > 800, (wrong, checked)
    assert int(ok.sum()) > 150
    worst = 0.0
    for s in np.nonzero(ok)[0]:
        ref = np.linalg.solve(H[s], g[s])
        cond = np.linalg.cond(H[s])
        err = np.abs(x[s] - ref).max() / max(np.abs(ref).max(), 1e-300)
        worst = max(worst, err / (cond * 2.2e-16))
        assert err <= 64 * cond * 2.2e-16, (s, kinds[s], err, cond)
    print("ldlt6: accepted", int(ok.sum()), "of", n, "; worst error / (cond * eps) =", worst)
    # 600 well-scaled systems (the Gauss-Newton normal equations of a scene that constrains all six degrees of freedom): every one accepted
    rng = np.random.default_rng(7)
    H2 = np.zeros((600, 6, 6)); g2 = np.zeros((600, 6))
    for s in range(600):
        J = rng.normal(size=(int(rng.integers(12, 2000)), 6)) * rng.uniform(0.05, 20.0, size=6)
        H2[s] = J.T @ J; g2[s] = -J.T @ (rng.normal(size=J.shape[0]) * 0.05)
    H2 = 0.5 * (H2 + H2.transpose(0, 2, 1))
    x2 = np.zeros((600, 6)); ok2 = np.zeros(600, dtype=np.int32)
    rc = _lib.lib().fls_debug_ldlt6(0, np.ascontiguousarray(H2.transpose(0, 2, 1)).reshape(600, 36).ctypes.data_as(dp), g2.ctypes.data_as(dp), 600,
                                    x2.ctypes.data_as(dp), ok2.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0 and ok2.all()
    for s in range(600):
        ref = np.linalg.solve(H2[s], g2[s])
        assert np.abs(x2[s] - ref).max() <= 64 * np.linalg.cond(H2[s]) * 2.2e-16 * np.abs(ref).max(), s

