"""Generates tests/golden/*.npz: small seeded inputs + the CPU oracle's outputs on them.

The reference has no golden vectors for the registration path and cannot be built here, so
these vectors pin the ORACLE (and, through the gpu tests, the HIP path) against silent drift and against
platform differences in the numpy scene generator: inputs are stored, not regenerated.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from funny_lidar_slam_amd import registration as reg, synth  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "p2plane_ivox": ("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, 1, 0.02, False),
    "icp_loc": ("IcpOptimized", reg.YAML_NCLT_ICP, 0, 0.15, True),
    "ndt": ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, 0.02, False),
    "loam_full": ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, 3, 0.03, False),
    "p2plane_kd_loc": ("PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, 1, 0.02, True),
}


def run_case(mode, y, cfg_id, scale, loc):
    cfg = synth.make_config(cfg_id, scale=scale)
    o = util.oracle_for(mode, y, loc)
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    o.AddCloudToLocalMap(*maps)
    corner = cfg.get("corner_scan")
    ok, T = o.Match(cfg["scan"], cfg["T_init"], src1=corner, update_map=False)
    Ts, nv, sr = o.iteration_log()
    out = dict(scan=cfg["scan"], map=cfg["map"], T_init=cfg["T_init"], T_gt=cfg["T_gt"], ok=np.array(ok), T=T, log_T=Ts, log_nv=nv,
               log_res=sr, iterations=np.array(o.stats.iterations), n_valid=np.array(o.stats.n_valid))
    ids, cnt, valid = o.correspondences(0)
    out.update(ids=ids, cnt=cnt, valid=valid)
    if corner is not None:
        ids1, cnt1, valid1 = o.correspondences(1)
        out.update(corner_scan=corner, corner_map=cfg["corner_map"], ids1=ids1, cnt1=cnt1, valid1=valid1,
                   n_valid_corner=np.array(o.stats.n_valid_corner))
    if loc:
        out["fitness2"] = np.array(o.GetFitnessScore(2.0), np.float32)
    return out


def run_sequence():
    """3-scan mapping replay with map updates (AddCloudToLocalMap rule, Q1/Q15 persistence, per-handle is_first)."""
    scene = synth.make_scene()
    rng = synth.rng_for(1, 7)
    radius = 22.0
    mp = synth.sample_map(scene, 40000, synth.rng_for(1, 0, 3), radius=radius)
    lid = dict(synth.VELODYNE_64, n_az=40)
    T = np.eye(4)
    scans, poses_gt = [], []
    for k in range(3):
        step = synth.random_pose(rng, 1.0, 0.25)
        T = T @ step
        poses_gt.append(T.copy())
        scans.append(synth.cast_scan(scene, T, rng=rng, max_range=radius, **lid))
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o.AddCloudToLocalMap(mp)
    guess = np.eye(4)
    out = dict(map=mp, scans=np.stack(scans), poses_gt=np.stack(poses_gt))
    Ts, oks, sizes, nvs, its = [], [], [], [], []
    for k in range(3):
        ok, Tk = o.Match(scans[k], guess, update_map=True)
        Ts.append(Tk); oks.append(ok); sizes.append(o.map_size()); nvs.append(o.stats.n_valid); its.append(o.stats.iterations)
        guess = Tk
    out.update(T=np.stack(Ts), ok=np.array(oks), map_sizes=np.array(sizes), n_valid=np.array(nvs), iterations=np.array(its))
    return out


FEAT16 = dict(vertical_scan=16, horizontal_scan=1800, min_distance=4.0, max_distance=100.0, corner_thres=1.0, planar_thres=0.1)


def run_features():
    """One Velodyne-16 frame (16 x 900 firings, lidar_model.cpp:24-30 columns) through the feature front-end oracle."""
    from oracle import oracle as O
    scene = synth.make_scene()
    raw = synth.cast_raw_scan(scene, synth.random_pose(synth.rng_for(0, 4), 10.0, 2.0), rng=synth.rng_for(0, 0, 4), **synth.VELODYNE_16)
    o = O.OracleFeatures(horizontal_resolution=float(np.float32(0.2) / 180.0 * np.pi), **FEAT16)
    o.Project(raw)
    o.ExtractFeatures()
    out = dict(raw=raw.view(np.uint8).reshape(raw.shape[0], raw.dtype.itemsize))
    for name in O.FEAT_ARRAYS:
        out[name] = o.get(name)
    return out


if __name__ == "__main__":
    for name, args in CASES.items():
        d = run_case(*args)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **d)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items() if k in ("scan", "map", "ids")}, "iters", int(d["iterations"]),
              "n_valid", int(d["n_valid"]), "ok", bool(d["ok"]))
    d = run_sequence()
    np.savez_compressed(os.path.join(HERE, "p2plane_ivox_sequence.npz"), **d)
    print("sequence", d["map_sizes"], d["n_valid"], d["iterations"], d["ok"])
    d = run_features()
    np.savez_compressed(os.path.join(HERE, "features_velodyne16.npz"), **d)
    print("features", {k: v.shape for k, v in d.items() if k in ("raw", "ordered", "corner", "planar")})
