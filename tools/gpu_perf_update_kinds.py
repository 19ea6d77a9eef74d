"""Mapping-mode cost per scan (Match + the map update the reference performs inside Match) for the non-iVox kinds,
GPU vs CPU oracle, on a short replay (BASELINE config sizes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from oracle import oracle as O
from tests import util

CASES = [(2, "IncrementalNDT", reg.YAML_NCLT_NDT), (0, "IcpOptimized", reg.YAML_NCLT_ICP), (3, "LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL)]
for cid, mode, y in CASES:
    if len(sys.argv) > 1 and str(cid) not in sys.argv[1:]:
        continue
    cfg = synth.make_config(cid)
    scene = cfg["scene"]
    lid = synth.VELODYNE_16 if cid == 0 else synth.VELODYNE_64
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    m = reg.make_matcher(mode, y)
    o = util.oracle_for(mode, y, False)
    O.set_threads(32)
    m.AddCloudToLocalMap(maps); o.AddCloudToLocalMap(*maps)
    rng = synth.rng_for(cid, 77)
    Tgt = np.eye(4); guess = np.eye(4)
    tg, tc = [], []
    for k in range(5):
        Tgt = Tgt @ synth.random_pose(rng, 0.5, 0.3)
        scan = synth.cast_scan(scene, Tgt, rng=rng, **lid)
        corner = synth.cast_edge_scan(scene, Tgt, 7680, rng) if cid == 3 else None
        if cid == 3:
            scan = scan[::2].copy()
        cl = util.cluster_for(mode, scan, corner)
        t = time.perf_counter(); m.UploadScan(cl); t_up = time.perf_counter() - t
        T = guess.copy(); t = time.perf_counter(); m.MatchResident(T, update_map=False); t_res = time.perf_counter() - t
        T = guess.copy(); t = time.perf_counter(); ok = m.Match(cl, T, update_map=True); tg.append(time.perf_counter() - t)
        print(f"     scan upload (incl. in-Match VoxelGrid) {1e3*t_up:.2f} ms, resident Match {1e3*t_res:.2f} ms, => map update ~{1e3*(tg[-1]-t_up-t_res):.2f} ms")
        t = time.perf_counter(); ok_ref, T_ref = o.Match(scan, guess, src1=corner, update_map=True); tc.append(time.perf_counter() - t)
        dt, dr = synth.pose_error(T, T_ref)
        print(f"  {mode} scan {k}: ok {ok}/{ok_ref} iters {m.stats.iterations}/{o.stats.iterations} GPU {1e3*tg[-1]:8.2f} ms  CPU(32 thr) {1e3*tc[-1]:8.2f} ms  pose diff {dt:.1e} m {dr:.1e} rad  map {m.map_size()}/{o.map_size()}", flush=True)
        guess = T_ref
    print(f"{mode}: median Match+update GPU {1e3*np.median(tg):.2f} ms, CPU oracle {1e3*np.median(tc):.2f} ms")
