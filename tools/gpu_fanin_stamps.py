"""Shader-clock stamps of the fit + Gauss-Newton tail of the iVox kind (BASELINE configs[1]) in the workgroup that runs the tail: ticket form vs
flag-in-data rows.  Needs the -DFLS_TIMING build: (cd funny_lidar_slam_amd/csrc && make timing), then
FLS_REG_LIB=funny_lidar_slam_amd/libfls_reg_timing.so python tools/gpu_fanin_stamps.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth, _lib
cfg = synth.make_config(1)
names = {0: "begin", 13: "fit done", 14: "wave sums done", 15: "row drained", 1: "tail start (ticket won / gatherer)", 2: "a-rows", 3: "rows reduced", 4: "solved", 5: "published"}
order = [13, 14, 15, 1, 3, 4, 5]
for arm in sys.argv[1:] or ["FLS_FANIN_LL=0", "FLS_FANIN_LL=1"]:
    os.environ.update(dict(kv.split("=", 1) for kv in arm.split()))
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX); m.AddCloudToLocalMap([cfg["map"]])
    cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"]); m.UploadScan(cl)
    run, Tv = m.resident_call(np.eye(4))
    for _ in range(20): run()
    ts = []
    for _ in range(100):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    rows = []
    for _ in range(15):
        run()
        st = (C.c_int64 * 16)()
        _lib.lib().fls_get_debug_stamps(m._h, st)
        v = list(st)
        rows.append([v[i] - v[0] for i in order])
    med = np.median(np.array(rows, dtype=np.float64), axis=0)
    print(f"[{arm}] match median {1e6*np.median(ts):.1f} us; ticks since the tail workgroup's begin (median of 15 Matches, last iteration): " +
          "  ".join(f"{names[i]}={int(x)}" for i, x in zip(order, med)), flush=True)
    m.close()
