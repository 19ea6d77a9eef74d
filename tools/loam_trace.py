import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util
cfg = synth.make_config(3)
m = reg.make_matcher("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL)
m.AddCloudToLocalMap([cfg["map"], cfg["corner_map"]])
cl = util.cluster_for("LoamFull_KdTree", cfg["scan"], cfg["corner_scan"])
m.UploadScan(cl)
for _ in range(10):
    T = np.eye(4); m.MatchResident(T)
print("iters", m.stats.iterations)
