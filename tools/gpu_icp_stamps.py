"""Shader-clock stamps of IcpOptimized's one-launch iteration (icp_knn_fit_kernel) in the workgroup that runs the Gauss-Newton tail.
Needs the -DFLS_TIMING build: (cd funny_lidar_slam_amd/csrc && make timing), then
FLS_REG_LIB=funny_lidar_slam_amd/libfls_reg_timing.so python tools/gpu_icp_stamps.py [icp|ndt]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth, _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "icp"
mode, y, cid, loc = {"icp": ("IcpOptimized", reg.YAML_NCLT_ICP, 0, True), "ndt": ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False)}[kind]
cfg = synth.make_config(cid)
names = {12: "search done (thread 0's wave)", 13: "fit + reduction done", 14: "workgroup barrier passed", 1: "row drained + ticket won: tail starts", 3: "rows read + reduced", 4: "solved", 5: "pose + state + mailbox issued"}
order = [12, 13, 14, 1, 3, 4, 5]
m = reg.make_matcher(mode, y, is_localization_mode=loc); m.AddCloudToLocalMap([cfg["map"]])
cl = reg.PointcloudCluster(ordered_cloud_=cfg["scan"]); m.UploadScan(cl)
run, Tv = m.resident_call(np.eye(4))
for _ in range(20): run()
ts = []
for _ in range(100):
    t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
rows = []
for _ in range(25):
    run()
    st = (C.c_int64 * 16)()
    _lib.lib().fls_get_debug_stamps(m._h, st)
    v = list(st)
    rows.append([v[i] - v[0] for i in order])
med = np.median(np.array(rows, dtype=np.float64), axis=0)
print(f"[{kind}] match median {1e6*np.median(ts):.1f} us, iterations {m.stats.iterations}; shader-clock ticks since the tail workgroup's entry (median of 25 Matches, last iteration):")
prev = 0
for i, x in zip(order, med):
    print(f"   {names[i]:42s} {int(x):7d}  (+{int(x - prev)})")
    prev = x
m.close()
