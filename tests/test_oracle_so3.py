"""Pins the oracle's SO3Hat / SO3Exp / RotationMatrixToRPY against the reference's OWN known-answer
tests (the only reference tests that touch the registration path: /root/reference/test/math_function_ut.cpp
:9-44 HatTest, :46-133 SO3ExpTest, :160-192 RotationMatrixToRPYTest).  Same inputs, same expected values."""
import numpy as np

from oracle import oracle as O


def hat_np(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def is_approx(a, b, prec=1e-12):  # Eigen isApprox: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)
    return np.sum((a - b) ** 2) <= prec * prec * min(np.sum(a * a), np.sum(b * b))


def test_hat_zero_input():  # HatTest.HandleZeroInput :9-14
    assert np.array_equal(O.so3_hat([0.0, 0.0, 0.0]), np.zeros((3, 3)))


def test_hat_123():  # HatTest.HandleIntInput / HandleFloatInput / HandleDoubleInput :16-44
    expect = np.array([[0.0, -3.0, 2.0], [3.0, 0.0, -1.0], [-2.0, 1.0, 0.0]])
    assert np.array_equal(O.so3_hat([1.0, 2.0, 3.0]), expect)


def test_exp_zero_is_identity():  # SO3ExpTest.HandleZeroInput :46-56
    assert np.array_equal(O.so3_exp([0.0, 0.0, 0.0]), np.eye(3))


def test_exp_pi_2():  # SO3ExpTest.HandleDoublePi2Input :82-91
    a = np.array([1.0, 0.0, 0.0])
    expect = np.outer(a, a) + 1.0 * hat_np(a)
    assert is_approx(O.so3_exp(a * np.pi / 2), expect)


def test_exp_pi_4():  # :93-104
    a = np.array([1.0, 0.0, 0.0])
    expect = np.cos(np.pi / 4) * np.eye(3) + (1 - np.cos(np.pi / 4)) * np.outer(a, a) + np.sin(np.pi / 4) * hat_np(a)
    assert is_approx(O.so3_exp(a * np.pi / 4), expect)


def test_exp_3pi_and_negative():  # :106-127
    a = np.array([1.0, 0.0, 0.0])
    expect = np.cos(np.pi) * np.eye(3) + (1 - np.cos(np.pi)) * np.outer(a, a) + np.sin(np.pi) * hat_np(a)
    assert is_approx(O.so3_exp(a * 3 * np.pi), expect)
    assert is_approx(O.so3_exp(-a * 3 * np.pi), expect)


def test_exp_transpose_is_inverse():  # SO3ExpTest.HandleTranspose :129-133
    a = np.array([1.0, 0.0, 0.0])
    assert is_approx(O.so3_exp(a).T, O.so3_exp(-a))


def test_exp_below_epsilon_is_identity():  # math_function.h:81: theta > epsilon() guard
    assert np.array_equal(O.so3_exp([1e-17, 0.0, 0.0]), np.eye(3))


def test_rpy_known_answer():  # RotationMatrixToRPYTest.HandleInput :160-192 (EXPECT_DOUBLE_EQ = 4 ulp)
    def rx(t): return np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])
    def ry(t): return np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
    def rz(t): return np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])
    R = rz(np.pi / 3) @ ry(np.pi / 4) @ rx(np.pi / 6)
    e = O.rpy(R)
    for got, want in zip(e, (np.pi / 6, np.pi / 4, np.pi / 3)):
        assert abs(got - want) <= 4 * np.spacing(want)


def test_exp_random_against_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for _ in range(50):
        v = rng.normal(size=3) * rng.uniform(1e-6, 3.0)
        assert np.allclose(O.so3_exp(v), Rotation.from_rotvec(v).as_matrix(), atol=1e-14)
