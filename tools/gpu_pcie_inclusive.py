"""fls_match (scan handed over as a host buffer: de-interleave + H2D inside the call) vs fls_match_resident on BASELINE configs[1]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(1)
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
m.AddCloudToLocalMap([cfg["map"]])
cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
cl8 = reg.PointcloudCluster(planar_cloud_=np.ascontiguousarray(np.pad(cfg["scan"], ((0, 0), (0, 5)))))  # pcl::PointXYZI stride
for name, c in (("packed xyz (12 B/pt)", cl), ("pcl::PointXYZI (32 B/pt)", cl8)):
    for _ in range(5):
        T = np.eye(4); m.Match(c, T, update_map=False)
    ts = []
    for _ in range(50):
        T = np.eye(4); t = time.perf_counter(); m.Match(c, T, update_map=False); ts.append(time.perf_counter() - t)
    print(f"fls_match, host buffer {name}: median {1e6*np.median(ts):.1f} us  -> {1/np.median(ts):.0f} scans/s")
m.UploadScan(cl)
ts = []
for _ in range(50):
    T = np.eye(4); t = time.perf_counter(); m.MatchResident(T); ts.append(time.perf_counter() - t)
print(f"fls_match_resident: median {1e6*np.median(ts):.1f} us -> {1/np.median(ts):.0f} scans/s")
