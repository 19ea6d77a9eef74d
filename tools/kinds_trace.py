"""Ten resident Matches of every kind at full BASELINE size (no oracle): workload for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util

for cid, mode, y, loc in ((0, "IcpOptimized", reg.YAML_NCLT_ICP, True), (1, "PointToPlane_IVOX", reg.YAML_NCLT_IVOX, False),
                          (2, "IncrementalNDT", reg.YAML_NCLT_NDT, False), (3, "LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, False),
                          (1, "PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, True)):
    cfg = synth.make_config(cid)
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    m.AddCloudToLocalMap(maps)
    m.UploadScan(util.cluster_for(mode, cfg["scan"], cfg.get("corner_scan")))
    for _ in range(10):
        T = np.eye(4); m.MatchResident(T)
    print(mode, "iterations", m.stats.iterations, "n_src", m.stats.n_source, flush=True)
    m.close()
