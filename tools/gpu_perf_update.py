"""Mapping-mode cost: Match + AddCloudToLocalMap (incremental device image) on BASELINE configs[1] size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(1)
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
t = time.perf_counter(); m.AddCloudToLocalMap([cfg["map"]]); print("initial map build ms", 1e3 * (time.perf_counter() - t))
scene = cfg["scene"]; rng = synth.rng_for(1, 123)
Tgt = cfg["T_gt"].copy(); guess = np.eye(4)
for k in range(6):
    scan = synth.cast_scan(scene, Tgt, rng=rng, **synth.VELODYNE_64)
    cl = reg.PointcloudCluster(planar_cloud_=scan)
    m.UploadScan(cl)
    T = guess.copy(); t = time.perf_counter(); ok = m.MatchResident(T, update_map=False); t_match0 = time.perf_counter() - t
    m.set_profiling(True)
    T = guess.copy(); t = time.perf_counter(); ok = m.MatchResident(T, update_map=False); t_match = time.perf_counter() - t
    ms, nl, pi = m.kernel_time(); m.set_profiling(False)
    print(f'   first call {1e6*t_match0:.1f} us; second call {1e6*t_match:.1f} us; knn kernel avg {1e3*ms/max(nl,1):.1f} us over {nl} launches')
    T = guess.copy(); t = time.perf_counter(); ok = m.MatchResident(T, update_map=True); t_both = time.perf_counter() - t
    print(f"scan {k}: match {1e6*t_match:8.1f} us, match+map update {1e6*t_both:9.1f} us, iters {m.stats.iterations}, map pts {m.map_size()}, voxels {m.map_size(102)}, "
          f"device updates {m.map_size(103)}, refused {m.map_size(104)}, host incremental {m.map_size(100)}, full rebuilds {m.map_size(101)}")
    guess = T
    Tgt = Tgt @ synth.random_pose(rng, 0.5, 0.5)
