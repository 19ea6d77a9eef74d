"""A/B of the IncrementalNDT correspondence kernel on BASELINE configs[2]: one lane per neighbour voxel (FLS_NDT_LANES=1, default) vs one
lane per point; any other switch of the kind can be an arm (one argument = the settings of one arm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(2)
arms = sys.argv[1:] or ["FLS_NDT_LANES=1", "FLS_NDT_LANES=0", "FLS_NDT_LANES=1"]   # e.g. "FLS_SOLVE_THREADS=256" "FLS_FUSED_TAIL=1"
for arm in arms:
    for k in ("FLS_NDT_LANES", "FLS_SOLVE_THREADS", "FLS_FUSED_TAIL"):
        os.environ.pop(k, None)
    os.environ.update(dict(kv.split("=", 1) for kv in arm.split()))
    m = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT); m.AddCloudToLocalMap([cfg["map"]])
    cl = reg.PointcloudCluster(ordered_cloud_=cfg["scan"]); m.UploadScan(cl)
    run, Tv = m.resident_call(np.eye(4))
    for _ in range(5): run()
    ts = []
    for _ in range(100):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(10): run()
    ms, nl, _ = m.kernel_time()
    m.set_profiling(False)
    print(f"[{arm}] match median {1e6*np.median(ts):.1f} us, iterations {m.stats.iterations}, n_valid {m.stats.n_valid}, correspondence launch {1e3*ms/max(nl,1):.2f} us, "
          f"T[0,3]={np.array(Tv).reshape(4,4)[0,3]:.15f}")
    m.close()
