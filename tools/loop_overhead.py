import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(1)
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX); m.AddCloudToLocalMap([cfg["map"]])
cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"]); m.UploadScan(cl)
run, Tv = m.resident_call(np.eye(4))
def step():
    rc = run()
    if rc < 0: raise RuntimeError
    return rc == 0, Tv
def timed(K, every, label):
    t0 = time.perf_counter()
    for k in range(K):
        if every and k % every == 0:
            m.set_profiling(True); ok, T = step(); m.set_profiling(False)
        else:
            ok, T = step()
    el = time.perf_counter() - t0
    print(f"{label}: K={K} every={every}: {1e6*el/K:.1f} us/step", flush=True)
for _ in range(5): step()
timed(20, 4, "cold W=5, events every 4th")
timed(20, 4, "again")
timed(50, 4, "K=50")
timed(50, 0, "K=50 no events")
timed(200, 0, "K=200 no events")
timed(200, 4, "K=200 events/4")
timed(200, 16, "K=200 events/16")
timed(20, 4, "K=20 events/4 (warm)")
time.sleep(0.5)
timed(20, 4, "K=20 after 0.5 s idle")
