"""The oracle's Eigen-style factorisations vs numpy/scipy (SURVEY.md 8c "how the oracle earns trust" item 3).
Eigen is not available in this container, so these third-party pieces are cross-checked numerically
(<= 1e-10 relative) rather than pinned bit-for-bit."""
import numpy as np
import pytest

from oracle import oracle as O


def test_colpiv_qr_5x3_matches_lstsq():
    rng = np.random.default_rng(2)
    for _ in range(200):
        A = rng.normal(size=(5, 3)) * rng.uniform(0.1, 50)
        b = -np.ones(5)
        x = O.colpiv_qr_solve_5x3(A, b)
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.allclose(x, ref, rtol=1e-10, atol=1e-12)


def test_colpiv_qr_plane_points_far_from_origin():
    rng = np.random.default_rng(3)
    n = np.array([0.2, -0.3, 0.93]); n /= np.linalg.norm(n)
    for _ in range(100):
        c = rng.uniform(-80, 80, 3)
        if abs(c @ n) < 5.0:  # A x = -1 cannot represent planes through the origin; keep the offset healthy
            c = c + n * (10.0 - c @ n)
        u = np.cross(n, [1, 0, 0]); u /= np.linalg.norm(u); v = np.cross(n, u)
        P = c + rng.uniform(-0.4, 0.4, (5, 1)) * u + rng.uniform(-0.4, 0.4, (5, 1)) * v + rng.normal(0, 0.01, (5, 1)) * n
        x = O.colpiv_qr_solve_5x3(P, -np.ones(5))
        ref = np.linalg.lstsq(P, -np.ones(5), rcond=None)[0]
        assert np.allclose(x, ref, rtol=1e-7, atol=1e-10)
        assert abs(abs(np.dot(x / np.linalg.norm(x), n)) - 1.0) < 5e-2


def test_colpiv_qr_rank_deficient_gives_basic_solution():
    # two identical columns: Eigen's solve zeroes the dropped pivot (basic, not minimum-norm solution)
    A = np.array([[1.0, 1.0, 0.0], [2.0, 2.0, 1.0], [3.0, 3.0, 0.5], [4.0, 4.0, -1.0], [5.0, 5.0, 2.0]])
    x = O.colpiv_qr_solve_5x3(A, -np.ones(5))
    assert np.count_nonzero(x == 0.0) >= 1
    r = A @ x + 1
    r_ref = A @ np.linalg.lstsq(A, -np.ones(5), rcond=None)[0] + 1
    assert np.linalg.norm(r) == pytest.approx(np.linalg.norm(r_ref), rel=1e-9)


def spd6(rng):
    J = rng.normal(size=(40, 6))
    return J.T @ J


def test_fullpiv_qr_6_matches_solve():
    rng = np.random.default_rng(4)
    for _ in range(100):
        H = spd6(rng); g = rng.normal(size=6)
        assert np.allclose(O.fullpiv_qr_solve_6(H, g), np.linalg.solve(H, g), rtol=1e-10, atol=1e-12)


def test_fullpiv_qr_6_rank_deficient_is_finite():
    # all normals parallel -> rank 1 H: Eigen truncates to rank() and zero-fills
    n = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 1.0])
    H = np.outer(n, n) * 7.0
    x = O.fullpiv_qr_solve_6(H, n * 3.0)
    assert np.all(np.isfinite(x))
    assert np.allclose(H @ x, n * 3.0, atol=1e-12)
    assert np.count_nonzero(x) == 1
    assert np.array_equal(O.fullpiv_qr_solve_6(np.zeros((6, 6)), np.ones(6)), np.zeros(6))


def test_lu_inverse_6():
    rng = np.random.default_rng(5)
    for _ in range(100):
        H = spd6(rng) + rng.normal(size=(6, 6)) * 0.1
        inv, det = O.lu_inverse_6(H)
        assert np.allclose(inv, np.linalg.inv(H), rtol=1e-9, atol=1e-12)
        assert det == pytest.approx(np.linalg.det(H), rel=1e-10)
    inv, det = O.lu_inverse_6(np.zeros((6, 6)))
    assert det == 0.0  # the exact-zero test icp_optimized.h:129 relies on


def test_inverse3_and_svd3():
    rng = np.random.default_rng(6)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        C = A @ A.T * rng.uniform(1e-4, 10)
        assert np.allclose(O.inverse3(C + 1e-3 * np.eye(3)), np.linalg.inv(C + 1e-3 * np.eye(3)), rtol=1e-8)
        U, S, V = O.svd3(C)
        assert np.all(np.diff(S) <= 0)
        assert np.allclose(S, np.linalg.svd(C, compute_uv=False), rtol=1e-10, atol=1e-14)
        assert np.allclose(U @ np.diag(S) @ V.T, C, rtol=1e-10, atol=1e-13)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
    # general (non-symmetric) input too
    A = rng.normal(size=(3, 3))
    U, S, V = O.svd3(A)
    assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12)


def test_svd3_line_direction():
    # 5 points along a line: s0 >> s1 and V[:,0] is the direction (CornerMatch gate, loam_full_kdtree.h:249-252)
    d = np.array([0.0, 0.0, 1.0])
    P = np.array([[10.0, 5.0, z] for z in (0.0, 0.3, 0.6, 0.9, 1.2)]) + 1e-3
    D = P - P.mean(0)
    U, S, V = O.svd3(D.T @ D / 5.0)
    assert S[0] > 3.0 * S[1]
    assert abs(abs(V[:, 0] @ d) - 1.0) < 1e-9
