"""A/B of two builds of libfls_reg.so on the resident Match of one BASELINE config: `python tools/gpu_ab_libs.py <cid> <libA> <libB> [reps]`.
Alternating sub-processes (A B A B), 200 timed Matches each, median + pose bits + iterations: a kernel change must leave the last two columns alone."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "one":
    import time
    import numpy as np
    from funny_lidar_slam_amd import registration as reg, synth
    from tests import util
    cid = int(sys.argv[2])
    mode, y, loc = {0: ("IcpOptimized", reg.YAML_NCLT_ICP, True), 1: ("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, False), 2: ("IncrementalNDT", reg.YAML_NCLT_NDT, False),
                    3: ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, False)}[cid]
    cfg = synth.make_config(cid)
    m = reg.make_matcher(mode, y, is_localization_mode=loc) if cid != 3 else reg.make_matcher(mode, y)
    m.AddCloudToLocalMap([cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else []))
    cl = util.cluster_for(mode, cfg["scan"], cfg.get("corner_scan"))
    m.UploadScan(cl)
    run, Tv = m.resident_call(cfg["T_init"] if cid == 1 else np.eye(4))
    for _ in range(40):
        run()
    ts = []
    for _ in range(200):
        t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
    T = np.array(Tv, dtype=np.float64)
    print(f"cid {cid} lib {os.path.basename(os.environ.get('FLS_REG_LIB', 'libfls_reg.so'))}: match median {1e6 * np.median(ts):.1f} us (min {1e6 * min(ts):.1f}), iterations {m.stats.iterations}, "
          f"n_valid {m.stats.n_valid}, pose sha {hash(T.tobytes()) & 0xffffffff:08x}", flush=True)
    m.close()
else:
    cid, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
    for lib in (a, b, a, b):
        env = dict(os.environ, FLS_REG_LIB=os.path.abspath(lib), PYTHONHASHSEED="0")
        p = subprocess.run([sys.executable, __file__, "one", cid], env=env, capture_output=True, text=True, timeout=300)
        print(p.stdout.strip() or p.stderr.strip()[-400:], flush=True)
