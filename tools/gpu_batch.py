"""Throughput of fls_match_batch (BASELINE configs[4] shape: independent 64x1800 scans against one 1e6-pt iVox map)
for several lane counts, on one GPU.  8 distinct scans are cycled to make the batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth

n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfgs = [synth.make_config(1, job=j) for j in range(8)]
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
m.AddCloudToLocalMap([cfgs[0]["map"]])
clusters = [reg.PointcloudCluster(planar_cloud_=cfgs[j % 8]["scan"]) for j in range(n_jobs)]
T0 = [np.eye(4)] * n_jobs
ref = None
for lanes in (8, 1, 2, 4, 8, 16):
    m.MatchBatch(clusters[:16], T0[:16], lanes=lanes)  # warm-up (lane creation, buffer growth)
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        oks, Ts, stats = m.MatchBatch(clusters, T0, lanes=lanes)
        ts.append(time.perf_counter() - t)
    best = min(ts)
    if ref is None:
        ref = Ts
    same = bool(np.array_equal(ref, Ts))
    print(f"lanes {lanes:2d}: {n_jobs} jobs, passes [ms] {[round(1e3*t,2) for t in ts]} -> best {n_jobs/best:8.1f} scans/s (scan upload included); "
          f"all ok {all(oks)}; iterations {sorted(set(s.iterations for s in stats))}; identical to first: {same}", flush=True)
