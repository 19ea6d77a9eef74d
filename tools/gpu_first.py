"""First GPU contact: smoke, full-size C2 parity vs oracle, rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
from funny_lidar_slam_amd import _lib, registration as reg, synth
from oracle import oracle as O

print("devices", _lib.device_count())
g.smoke()
cfg = synth.make_config(1)
y = reg.YAML_NCLT_IVOX
m = reg.make_matcher("PointToPlane_IVOX", y)
t = time.time(); m.AddCloudToLocalMap([cfg["map"]]); print("gpu add map", time.time() - t)
cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
T = cfg["T_init"].copy()
t = time.time(); ok = m.Match(cl, T, update_map=False); print("gpu match (cold)", time.time() - t, ok, m.stats.iterations, m.stats.n_valid, m.stats.sum_res)
o = O.OracleMatcher(O.P2PLANE_IVOX, O.Params(max_iterations=10, point_to_planar_thres=0.1, position_converge_thres=0.005, rotation_converge_thres=0.001))
o.AddCloudToLocalMap(cfg["map"])
t = time.time(); ok_ref, T_ref = o.Match(cfg["scan"], cfg["T_init"], update_map=False); print("oracle match", time.time() - t, ok_ref, o.stats.iterations, o.stats.n_valid, o.stats.sum_res, "threads", O.get_threads())
print("pose err gpu-vs-oracle", synth.pose_error(T, T_ref), "gpu-vs-gt", synth.pose_error(T, cfg["T_gt"]))
Tg, nvg, srg = m.iteration_log(); To, nvo, sro = o.iteration_log()
for i in range(len(nvo)):
    print(i, nvg[i], nvo[i], srg[i], sro[i], synth.pose_error(Tg[i], To[i]))
ids, cnt, valid = m.correspondences(); ids_r, cnt_r, valid_r = o.correspondences()
print("cnt equal", np.array_equal(cnt, cnt_r), "valid equal", np.array_equal(valid, valid_r), "slot0 equal", np.array_equal(ids[:, 0], ids_r[:, 0]),
      "sets equal", np.array_equal(np.sort(ids, 1), np.sort(ids_r, 1)), "mismatch rows", int((np.sort(ids, 1) != np.sort(ids_r, 1)).any(1).sum()))
c = o.counters(); print("oracle counters", c.point_iters, c.probes, c.hit_voxels, c.cand_points, c.tie_queries)
# timing, resident
m2 = reg.make_matcher("PointToPlane_IVOX", y); m2.AddCloudToLocalMap([cfg["map"]]); m2.UploadScan(cl)
for _ in range(3):
    T = cfg["T_init"].copy(); m2.MatchResident(T)
ts = []
for _ in range(20):
    m2b = None
    T = cfg["T_init"].copy(); t = time.perf_counter(); m2.MatchResident(T); ts.append(time.perf_counter() - t)
print("resident match ms: median %.3f min %.3f" % (1e3 * np.median(ts), 1e3 * min(ts)), "iters", m2.stats.iterations)
m2.set_profiling(True)
for _ in range(5):
    T = cfg["T_init"].copy(); m2.MatchResident(T)
ms, nl, pi = m2.kernel_time(); print("profiled corr kernel: total ms %.3f launches %d avg us %.2f point_iters %d" % (ms, nl, 1e3 * ms / max(nl, 1), pi))
print("device traffic counters", m2.traffic_counters())
