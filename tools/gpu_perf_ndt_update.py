"""Incremental-NDT in mapping mode (Match + the map update the reference performs inside Match): ms per scan and, with
FLS_HOST_TIMING=1, the host-side split of every update (stderr)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funny_lidar_slam_amd import registration as reg, synth  # noqa: E402

cfg = synth.make_config(2)
scene = cfg["scene"]
rng = synth.rng_for(2, 321)
Tgt = cfg["T_gt"].copy()
m = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT)
t = time.perf_counter(); m.AddCloudToLocalMap([cfg["map"]]); print("initial map (1e6 pts): %.1f ms" % (1e3 * (time.perf_counter() - t)))
guess = np.eye(4)
for k in range(6):
    scan = synth.cast_scan(scene, Tgt, rng=rng, **synth.VELODYNE_64)
    cl = reg.PointcloudCluster(ordered_cloud_=scan)
    T = guess.copy(); t = time.perf_counter(); ok = m.Match(cl, T, update_map=False); t_m = time.perf_counter() - t
    T = guess.copy(); t = time.perf_counter(); ok = m.Match(cl, T, update_map=True); t_b = time.perf_counter() - t
    print(f"scan {k}: Match {1e3 * t_m:.3f} ms, Match + update {1e3 * t_b:.3f} ms, iterations {m.stats.iterations}, voxels {m.map_size()}")
    guess = T
    Tgt = Tgt @ synth.random_pose(rng, 0.3, 0.2)
m.close()
