"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU part: the oracle reproduces its own stored outputs bit-for-bit on the stored inputs (pins the checker).
GPU part: the HIP path, fed the same stored inputs, matches the stored oracle outputs to the parity bar."""
import os

import numpy as np
import pytest

from funny_lidar_slam_amd import registration as reg, synth
from tests import util
from tests.golden.make_golden import CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, f"{name}.npz")))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(name):
    mode, y, _, _, loc = CASES[name]
    g = load(name)
    o = util.oracle_for(mode, y, loc)
    maps = [g["map"]] + ([g["corner_map"]] if "corner_map" in g else [])
    o.AddCloudToLocalMap(*maps)
    ok, T = o.Match(g["scan"], g["T_init"], src1=g.get("corner_scan"), update_map=False)
    Ts, nv, sr = o.iteration_log()
    assert ok == bool(g["ok"]) and o.stats.iterations == int(g["iterations"])
    assert np.array_equal(nv, g["log_nv"])
    assert np.allclose(Ts, g["log_T"], rtol=0, atol=1e-12)   # libm sin/cos may differ in the last ulp across hosts
    assert np.allclose(sr, g["log_res"], rtol=1e-12)
    ids, cnt, valid = o.correspondences(0)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(cnt, g["cnt"]) and np.array_equal(valid, g["valid"])
    if "ids1" in g:
        ids1, cnt1, valid1 = o.correspondences(1)
        assert np.array_equal(ids1, g["ids1"]) and np.array_equal(valid1, g["valid1"])
    if loc:
        assert o.GetFitnessScore(2.0) == pytest.approx(float(g["fitness2"]), rel=1e-6)


def test_oracle_reproduces_golden_sequence():
    g = load("p2plane_ivox_sequence")
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o.AddCloudToLocalMap(g["map"])
    guess = np.eye(4)
    for k in range(g["scans"].shape[0]):
        ok, T = o.Match(g["scans"][k], guess, update_map=True)
        assert ok == bool(g["ok"][k]) and o.stats.iterations == int(g["iterations"][k]) and o.stats.n_valid == int(g["n_valid"][k])
        assert o.map_size() == int(g["map_sizes"][k])
        assert np.allclose(T, g["T"][k], atol=1e-12)
        guess = T


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_matches_golden(name, built):
    mode, y, _, _, loc = CASES[name]
    g = load(name)
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    maps = [g["map"]] + ([g["corner_map"]] if "corner_map" in g else [])
    m.AddCloudToLocalMap(maps)
    T = g["T_init"].copy()
    ok = m.Match(util.cluster_for(mode, g["scan"], g.get("corner_scan")), T, update_map=False)
    Ts, nv, sr = m.iteration_log()
    assert ok == bool(g["ok"]) and m.stats.iterations == int(g["iterations"])
    assert np.array_equal(nv, g["log_nv"])
    for i in range(len(nv)):
        good, dt, dr = util.pose_close(Ts[i], g["log_T"][i])
        assert good, (i, dt, dr)
    ids, cnt, valid = m.correspondences(0)
    assert np.array_equal(cnt, g["cnt"]) and np.array_equal(valid, g["valid"])
    if mode == "PointToPlane_IVOX":  # slots 1..4 are introselect-ordered in the reference: compare as sets
        assert np.array_equal(ids[:, 0], g["ids"][:, 0]) and np.array_equal(np.sort(ids, 1), np.sort(g["ids"], 1))
    else:
        assert np.array_equal(ids, g["ids"])
    if "ids1" in g:
        ids1, cnt1, valid1 = m.correspondences(1)
        assert np.array_equal(ids1, g["ids1"]) and np.array_equal(valid1, g["valid1"])
        assert m.stats.n_valid_corner == int(g["n_valid_corner"])
    if loc:
        assert m.GetFitnessScore(2.0) == pytest.approx(float(g["fitness2"]), rel=1e-6)


@pytest.mark.gpu
def test_hip_matches_golden_sequence(built):
    """Mapping-mode replay: Match -> AddCloudToLocalMap rule -> next Match, 3 scans (map growth must be identical)."""
    g = load("p2plane_ivox_sequence")
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([g["map"]])
    guess = np.eye(4)
    for k in range(g["scans"].shape[0]):
        T = guess.copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=g["scans"][k]), T, update_map=True)
        assert ok == bool(g["ok"][k]) and m.stats.iterations == int(g["iterations"][k]) and m.stats.n_valid == int(g["n_valid"][k])
        assert m.map_size() == int(g["map_sizes"][k])
        good, dt, dr = util.pose_close(T, g["T"][k])
        assert good, (k, dt, dr)
        guess = T


def _features_raw(g):
    return np.ascontiguousarray(g["raw"]).view(synth.RAW_POINT_DTYPE).reshape(-1)


def test_oracle_reproduces_golden_features():
    from oracle import oracle as O
    from tests.golden.make_golden import FEAT16
    g = load("features_velodyne16")
    o = O.OracleFeatures(horizontal_resolution=float(np.float32(0.2) / 180.0 * np.pi), **FEAT16)
    o.Project(_features_raw(g))
    assert o.ExtractFeatures()
    for name in O.FEAT_ARRAYS:
        assert np.array_equal(o.get(name).view(np.uint8), g[name].view(np.uint8)), name


@pytest.mark.gpu
def test_gpu_matches_golden_features(built):
    from funny_lidar_slam_amd import features
    from oracle import oracle as O
    from tests.golden.make_golden import FEAT16
    g = load("features_velodyne16")
    f = features.FeatureFrontEnd(FEAT16["horizontal_scan"], FEAT16["vertical_scan"], float(np.float32(0.2) / 180.0 * np.pi), FEAT16["min_distance"],
                                 FEAT16["max_distance"], FEAT16["corner_thres"], FEAT16["planar_thres"])
    f.project(_features_raw(g))
    f.extract()
    for name in O.FEAT_ARRAYS:
        assert np.array_equal(f.get(name).view(np.uint8), g[name].view(np.uint8)), name
