"""A/B of the IcpOptimized iteration on BASELINE configs[0]: fused search + fit launch (FLS_ICP_FUSED=1, default) vs separate launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(0)
for fused in ("1", "0", "1"):
    os.environ["FLS_ICP_FUSED"] = fused
    m = reg.make_matcher("IcpOptimized", reg.YAML_NCLT_ICP, is_localization_mode=True); m.AddCloudToLocalMap([cfg["map"]])
    cl = reg.PointcloudCluster(ordered_cloud_=cfg["scan"]); m.UploadScan(cl)
    run, Tv = m.resident_call(np.eye(4))
    for _ in range(5): run()
    ts = []
    for _ in range(60):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    print(f"FLS_ICP_FUSED={fused}: match median {1e6*np.median(ts):.1f} us, iterations {m.stats.iterations}, T[0,3]={np.array(Tv).reshape(4,4)[0,3]:.15f}")
    m.close()
