"""BASELINE configs[4] through the NATIVE one-process / N-device path (fls_replicas_*: include/fls_reg.h, csrc/replicas.hpp):
the owner's map image replicated per device, 512 jobs block-partitioned over the devices, one host thread per device inside the
library.  usage: python tools/gpu_replicas.py [n_jobs] [device list, e.g. 0,1,2,3,4,5,6,7 | default: every visible device]
On a one-GPU box a list like 0,0 puts two handles on the one device (what tests/test_gpu_replicas.py does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib, registration as reg, synth

n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
devices = [int(d) for d in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(_lib.lib().fls_device_count()))
cfgs = [synth.make_config(1, job=j) for j in range(8)]
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, device_id=devices[0])
m.AddCloudToLocalMap([cfgs[0]["map"]])
clusters = [reg.PointcloudCluster(planar_cloud_=cfgs[j % 8]["scan"]) for j in range(n_jobs)]
T0 = [np.eye(4)] * n_jobs
oks0, Ts0, st0 = m.MatchBatch(clusters, T0, lanes=8)
t = time.perf_counter(); rs = m.Replicas(devices); t_rep = time.perf_counter() - t
print(f"devices {devices}: replication {1e3*t_rep:.1f} ms wall, import per entry [ms] {[round(x, 1) for x in rs.import_ms()]}", flush=True)
for lanes in (8, 16):
    rs.MatchBatch(clusters[:4 * len(devices)], T0[:4 * len(devices)], lanes=lanes)  # warm-up (lane creation, buffer growth)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); oks, Ts, stats = rs.MatchBatch(clusters, T0, lanes=lanes); ts.append(time.perf_counter() - t)
    print(f"lanes/device {lanes:2d}: {n_jobs} jobs over {len(devices)} entries, passes [ms] {[round(1e3*x, 2) for x in ts]} -> best {n_jobs/min(ts):9.1f} scans/s "
          f"(scan uploads included); identical to the owner's serial batch: {bool(np.array_equal(Ts, Ts0))}", flush=True)
