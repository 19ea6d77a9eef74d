"""One-off stress: many random scans per kind against the oracle with the full parity assertions of the test-suite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import test_gpu_parity as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
job0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100  # first synth job (seed): a second run with another offset sees other scans
fails = 0
t0 = time.time()
for job in range(job0, job0 + n):
    for mode, y, cid, scale, loc in (("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, 1, 0.04, False), ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, 0.04, False),
                                     ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, 3, 0.04, False), ("IcpOptimized", reg.YAML_NCLT_ICP, 0, 1.0, True),
                                     ("PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, 1, 0.04, True)):
        cfg = synth.make_config(cid, job=job, scale=scale)
        maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
        try:
            m, o, T, T_ref = tp.run_pair(mode, y, maps, cfg["scan"], corner=cfg.get("corner_scan"), loc=loc, sets_only_tail=(mode == "PointToPlane_IVOX"))
            m.close(); o.close()
        except AssertionError as e:
            fails += 1
            print("FAIL", mode, job, str(e)[:300], flush=True)
print(f"{n} scans (jobs {job0}..{job0 + n - 1}) x 5 kinds: {fails} failures, {time.time()-t0:.0f} s")
