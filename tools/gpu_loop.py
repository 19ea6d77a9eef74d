"""fls_loop_match on the synthetic sub-map pair of bench.py's loop_closure leg: exact host filters (default) and device filters
(FLS_DEVICE_VOXELGRID=1), next to the CPU oracle; workload for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from oracle import oracle as O
from tests import loopdata
src, tgt, Tt = loopdata.make_pair(job=1, n_az=450, n_t=5, n_s=3)
t = time.perf_counter(); fo, To, so = O.loop_match(src, tgt, np.eye(4)); t_cpu = time.perf_counter() - t
for env in ("0", "1"):
    os.environ["FLS_DEVICE_VOXELGRID"] = env
    ts = []
    for k in range(5):
        T = np.eye(4); t = time.perf_counter(); f, st = reg.LoopClosureMatch(src, tgt, T); ts.append(time.perf_counter() - t)
    print(f"FLS_DEVICE_VOXELGRID={env}: {1e3*np.median(ts[1:]):.2f} ms per Match (oracle {1e3*t_cpu:.1f} ms); fitness {f:.6f} vs {fo:.6f}; vs oracle {synth.pose_error(T, To)}; "
          f"vs truth {synth.pose_error(T, Tt)}; NDT evals {list(st.ndt_evaluations)}, GICP {st.gicp_iterations}/{st.gicp_inner_iterations}/{st.gicp_evaluations}")
