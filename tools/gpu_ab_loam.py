"""A/B of the LOAM correspondence path on BASELINE configs[3]: parity against the oracle, resident Match time, correspondence launch time.
usage: python tools/gpu_ab_loam.py "FLS_LOAM_DUAL=1" "FLS_LOAM_DUAL=0" "FLS_GRID27=1 FLS_LOAM_DUAL=0"   (one argument = the settings of one arm)
FLS_LOAM_DUAL: both feature classes in one correspondence + one fit launch (default 1); FLS_GRID27: one-stage 27-cell kernel on
gate-sized cells instead of the nearest-first two-stage kernel on half-gate cells (default 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util

cfg = synth.make_config(3)
y = reg.YAML_NCLT_LOAM_FULL
o = util.oracle_for("LoamFull_KdTree", y); o.AddCloudToLocalMap(cfg["map"], cfg["corner_map"])
ok_ref, T_ref = o.Match(cfg["scan"], np.eye(4), src1=cfg["corner_scan"], update_map=False)
cl = util.cluster_for("LoamFull_KdTree", cfg["scan"], cfg["corner_scan"])
for arm in (sys.argv[1:] or ["FLS_LOAM_DUAL=1", "FLS_LOAM_DUAL=0"]):
    for k in ("FLS_LOAM_DUAL", "FLS_GRID27", "FLS_FUSED_TAIL"):
        os.environ.pop(k, None)
    os.environ.update(dict(kv.split("=", 1) for kv in arm.split()))
    m = reg.make_matcher("LoamFull_KdTree", y); m.AddCloudToLocalMap([cfg["map"], cfg["corner_map"]])
    T = np.eye(4); ok = m.Match(cl, T, update_map=False)
    try:
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, slots=(0, 1), max_tie_rows=int(o.counters().tie_queries))
        par = "parity OK"
    except AssertionError as e:
        par = "PARITY FAIL " + str(e)[:200]
    m.UploadScan(cl)
    run, Tv = m.resident_call(np.eye(4))
    for _ in range(5):
        run()
    ts = []
    for _ in range(50):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(10):
        run()
    ms, nl, pi = m.kernel_time()
    m.set_profiling(False)
    print(f"[{arm}] {par}; iters {m.stats.iterations}; match median {1e6*np.median(ts):7.1f} us  min {1e6*min(ts):7.1f};"
          f" correspondence launches (corner + surf kNN + fits) avg {1e3*ms/max(nl,1):6.2f} us / {nl}", flush=True)
    m.close()
