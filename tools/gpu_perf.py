"""A/B timing of the iVox correspondence-kernel variants on the full BASELINE configs[1] workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util

cfg = synth.make_config(1)
y = reg.YAML_NCLT_IVOX
o = util.oracle_for("PointToPlane_IVOX", y); o.AddCloudToLocalMap(cfg["map"])
ok_ref, T_ref = o.Match(cfg["scan"], np.eye(4), update_map=False)
cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
variants = sys.argv[1:] or ["4", "8"]
for v in variants:
    os.environ["FLS_IVOX_VARIANT"] = v
    m = reg.make_matcher("PointToPlane_IVOX", y); m.AddCloudToLocalMap([cfg["map"]])
    T = np.eye(4); ok = m.Match(cl, T, update_map=False)
    try:
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        par = "parity OK"
    except AssertionError as e:
        par = "PARITY FAIL " + str(e)[:200]
    m.UploadScan(cl)
    for _ in range(5):
        T = np.eye(4); m.MatchResident(T)
    ts = []
    for _ in range(30):
        T = np.eye(4); t = time.perf_counter(); m.MatchResident(T); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(10):
        T = np.eye(4); m.MatchResident(T)
    ms, nl, pi = m.kernel_time()
    print(f"variant {v:>2}: {par}; iters {m.stats.iterations}; match median {1e6*np.median(ts):8.1f} us  min {1e6*min(ts):8.1f} us;"
          f" corr kernel avg {1e3*ms/max(nl,1):7.2f} us over {nl} launches", flush=True)
    if os.environ.get("FLS_DUMP_DBG"):
        import ctypes as C
        from funny_lidar_slam_amd import _lib
        st = (C.c_int64 * 16)()
        _lib.lib().fls_get_debug_stamps(m._h, st)
        v = list(st)
        print("   solve-kernel cycle stamps (delta from start):", [v[i] - v[0] for i in range(15)])
    m.close()
