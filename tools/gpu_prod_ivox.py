"""The adapter's real call for the default mode: fls_match from HOST buffers with update_map = 1 on the 0.5 m voxel-filtered planar cloud the
production pipeline feeds (preprocessing.cpp:236-237), a run of consecutive scans (0.5 m / 0.5 deg steps).  Prints per-scan wall time and
iterations; FLS_HOST_TIMING=1 adds the host-side split of every map update; workload for rocprofv3 --kernel-trace.
usage: python tools/gpu_prod_ivox.py [n_scans]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfg = synth.make_config(1)
rng = synth.rng_for(1, 123)
Tgt = cfg["T_gt"].copy()
scans = []
for k in range(n_scans):
    scans.append(synth.cast_scan(cfg["scene"], Tgt, rng=rng, **synth.VELODYNE_64))
    Tgt = Tgt @ synth.random_pose(rng, 0.5, 0.5)
filt = lambda c: np.ascontiguousarray(reg.VoxelGridCloud(c, 0.5)[:, :3])
fscans = [filt(s) for s in scans]
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
m.AddCloudToLocalMap([filt(cfg["map"])])
guess = np.eye(4)
tb, its = [], []
for sc in fscans:
    T = guess.copy(); t = time.perf_counter(); m.Match(reg.PointcloudCluster(planar_cloud_=sc), T, update_map=True); tb.append(time.perf_counter() - t)
    its.append(int(m.stats.iterations))
    guess = T
tb = np.array(tb[2:])
print(f"{n_scans} scans of ~{int(np.median([s.shape[0] for s in fscans]))} points: Match + map update from host buffers median {1e3*np.median(tb):.3f} ms, p10 {1e3*np.percentile(tb,10):.3f}, "
      f"max {1e3*tb.max():.3f}; iterations {its}; device batches {m.map_size(103)}, refused {m.map_size(104)}, map points {m.map_size()}")
