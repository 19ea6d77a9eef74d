"""Timing of every BASELINE config (full size) on the GPU next to the CPU oracle: rows of BASELINE.md section 4."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from oracle import oracle as O
from tests import util

CASES = [
    (0, "IcpOptimized", reg.YAML_NCLT_ICP, True),
    (1, "PointToPlane_IVOX", reg.YAML_NCLT_IVOX, False),
    (2, "IncrementalNDT", reg.YAML_NCLT_NDT, False),
    (3, "LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, False),
]
rows = []
for cid, mode, y, loc in CASES:
    if len(sys.argv) > 1 and str(cid) not in sys.argv[1:]:
        continue
    cfg = synth.make_config(cid)
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    corner = cfg.get("corner_scan")
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    t = time.perf_counter(); m.AddCloudToLocalMap(maps); t_add = time.perf_counter() - t
    cl = util.cluster_for(mode, cfg["scan"], corner)
    T = np.eye(4); ok = m.Match(cl, T, update_map=False)
    m.UploadScan(cl)
    for _ in range(3):
        T = np.eye(4); m.MatchResident(T)
    ts = []
    for _ in range(20):
        T = np.eye(4); t = time.perf_counter(); m.MatchResident(T); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(5):
        T = np.eye(4); m.MatchResident(T)
    ms, nl, pi = m.kernel_time()
    # CPU oracle, best of a few thread counts
    best = None
    one_thread = None
    for thr in (1, 16, 64):
        O.set_threads(thr)
        o = util.oracle_for(mode, y, loc); o.AddCloudToLocalMap(*maps)
        tt = []
        for _ in range(3):
            t = time.perf_counter(); ok_ref, T_ref = o.Match(cfg["scan"], np.eye(4), src1=corner, update_map=False); tt.append(time.perf_counter() - t)
        if thr == 1:
            one_thread = min(tt)
        if best is None or min(tt) < best[0]:
            best = (min(tt), thr, o.stats.iterations, o.stats.n_valid)
    dt, dr = synth.pose_error(T, T_ref)
    row = dict(config=cid, mode=mode, n_src=int(m.stats.n_source), n_src_corner=int(m.stats.n_source_corner), map=int(cfg["map"].shape[0]),
               iters=int(m.stats.iterations), ok=bool(ok), gpu_match_us=1e6 * float(np.median(ts)), gpu_scans_s=1.0 / float(np.median(ts)),
               corr_kernel_avg_us=1e3 * ms / max(nl, 1), cpu_match_ms=1e3 * best[0], cpu_threads=best[1], cpu_scans_s=1.0 / best[0], cpu_1thr_scans_s=1.0 / one_thread,
               speedup=best[0] / float(np.median(ts)), pose_err_m=dt, pose_err_rad=dr, add_map_ms=1e3 * t_add)
    rows.append(row)
    print(json.dumps(row), flush=True)
