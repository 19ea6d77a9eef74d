"""What an at-capacity map update costs with the evictions on the device (default) and with the round-2 rule (every batch that would
evict goes to the exact host code: FLS_IVOX_DEVICE_EVICT=0 / FLS_NDT_DEVICE_EVICT=0).  The scenarios are the ones of the eviction tests
(tests/test_gpu_parity.py::test_mapping_replay_device_evictions_straight_run, tests/test_gpu_mapping_replay.py::..._with_conflicts):
this tool only times them, the tests hold the parity assertions.  Usage: python tools/gpu_evict_timing.py  (one GPU, ~40 s)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_ivox():
    from funny_lidar_slam_amd import registration as reg, synth
    from tests import replay
    cap = 5000
    start = np.eye(4)
    start[1, 3] = 18.0
    r = replay.make_replay("ivox", n_frames=40, yaw_long_deg=0.0, start=start, max_range=20.0)
    scene = synth.make_scene()
    s0 = synth.cast_scan(scene, start, rng=synth.rng_for(5, 99), max_range=20.0, **dict(synth.VELODYNE_64, n_az=600))
    init = (s0.astype(np.float64) @ start[:3, :3].T + start[:3, 3]).astype(np.float32)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([init])
    Tp = start.copy()
    ts = []
    for f in r["frames"]:
        T = Tp @ f["guess_step"]
        cl = reg.PointcloudCluster(planar_cloud_=f["scan"])
        t = time.perf_counter()
        m.Match(cl, T, update_map=True)
        ts.append((time.perf_counter() - t, m.map_size(102) == cap - 1))
        Tp = T
    at_cap = [1e3 * t for t, c in ts if c]
    out = dict(kind="ivox", capacity=cap, frames=len(ts), frames_at_capacity=len(at_cap), ms_match_plus_update_at_capacity_median=float(np.median(at_cap)) if at_cap else None,
               device_batches=m.map_size(103), refused=m.map_size(104), evicted_on_device=m.map_size(117), recreated_on_device=m.map_size(126))
    m.close()
    return out


def run_ndt():
    from funny_lidar_slam_amd import registration as reg
    from tests import replay, util
    r = replay.make_replay("ndt")
    y = r["y"]
    m = reg.make_matcher("IncrementalNDT", y)
    m.AddCloudToLocalMap(r["init_clouds"])
    Tp = np.eye(4)
    ts = []
    for f in r["frames"]:
        T = Tp @ f["guess_step"]
        cl = util.cluster_for("IncrementalNDT", f["scan"], f["corner"])
        t = time.perf_counter()
        m.Match(cl, T, update_map=True)
        ts.append((time.perf_counter() - t, m.map_size(0) == y["ndt_capacity"] - 1))
        Tp = T
    at_cap = [1e3 * t for t, c in ts if c]
    out = dict(kind="ndt", capacity=y["ndt_capacity"], frames=len(ts), frames_at_capacity=len(at_cap),
               ms_match_plus_update_at_capacity_median=float(np.median(at_cap)) if at_cap else None,
               device_batches=m.map_size(109), refused=m.map_size(110), evicted_on_device=m.map_size(117), recreated_on_device=m.map_size(126))
    m.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(json.dumps(run_ivox() if sys.argv[1] == "ivox" else run_ndt()))
        sys.exit(0)
    for kind, var in (("ivox", "FLS_IVOX_DEVICE_EVICT"), ("ndt", "FLS_NDT_DEVICE_EVICT")):
        for val in ("1", "0"):
            env = dict(os.environ, **{var: val, "FLS_IVOX_CAPACITY": "5000"})
            p = subprocess.run([sys.executable, os.path.abspath(__file__), kind], env=env, capture_output=True, text=True, timeout=200)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
            print(f"{var}={val}: {line}", flush=True)
