"""A/B timing of environment-switch variants on the full BASELINE configs[1] workload, with the parity check of the test suite.
usage: python tools/gpu_ab.py "FLS_IVOX_FIT=1" "FLS_IVOX_FIT=2" ...   (each argument: space-separated VAR=VALUE settings of one arm)
With FLS_DUMP_DBG=1 and FLS_REG_LIB=<-DFLS_TIMING build> the shader-clock stamps of the fit / Gauss-Newton tail phases are printed."""
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
for arm in (sys.argv[1:] or [""]):
    sets = dict(kv.split("=", 1) for kv in arm.split()) if arm else {}
    old = {k: os.environ.get(k) for k in sets}
    os.environ.update(sets)
    m = reg.make_matcher("PointToPlane_IVOX", y); m.AddCloudToLocalMap([cfg["map"]])
    T = np.eye(4); ok = m.Match(cl, T, update_map=False)
    try:
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        par = "parity OK"
    except AssertionError as e:
        par = "PARITY FAIL " + str(e)[:200]
    m.UploadScan(cl)
    run, Tv = m.resident_call(np.eye(4))
    for _ in range(10):
        run()
    ts = []
    for _ in range(200):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(20):
        run()
    ms, nl, pi = m.kernel_time()
    m.set_profiling(False)
    print(f"[{arm or 'default':40s}] {par}; iters {m.stats.iterations}; match median {1e6*np.median(ts):7.1f} us  p10 {1e6*np.percentile(ts,10):7.1f}  min {1e6*min(ts):7.1f};"
          f" kNN kernel avg {1e3*ms/max(nl,1):6.2f} us / {nl} launches", flush=True)
    if os.environ.get("FLS_DUMP_DBG"):
        import ctypes as C
        from funny_lidar_slam_amd import _lib
        st = (C.c_int64 * 16)()
        _lib.lib().fls_get_debug_stamps(m._h, st)
        v = list(st)
        names = {0: "fit begin", 13: "fit end", 14: "reduce+ticket end", 1: "tail start", 2: "a-rows", 3: "b-rows reduced", 6: "qr pivot0", 7: "qr lds0", 8: "qr sel0", 9: "qr refl0",
                 10: "qr upd0", 11: "qr factor end", 12: "qr solve end", 4: "after qr", 5: "published"}
        order = [0, 13, 14, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 4, 5]
        print("   last-workgroup cycle stamps (100 MHz-independent shader clock ticks since fit begin):")
        base0 = v[0] if v[0] else v[1]
        print("   " + "  ".join(f"{names[i]}={v[i]-base0}" for i in order if v[i]))
    m.close()
    for k, vv in old.items():
        if vv is None: os.environ.pop(k, None)
        else: os.environ[k] = vv
