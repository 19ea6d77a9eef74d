"""What the source VoxelGrid costs inside fls_match for the ICP / NDT kinds: host filter (0), device filter with std::sort's order
(1, default since round 4: bit-identical), device filter with the stable radix sort (2, round-2/3 contract).  One process per setting.
usage: python tools/gpu_perf_voxelgrid.py            (spawns the four runs)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(kind):
    from funny_lidar_slam_amd import registration as reg, synth
    mode, y, cid, loc = {"icp": ("IcpOptimized", reg.YAML_NCLT_ICP, 0, True), "ndt": ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False)}[kind]
    cfg = synth.make_config(cid)
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    m.AddCloudToLocalMap([cfg["map"]])
    cl = reg.PointcloudCluster(ordered_cloud_=cfg["scan"])
    def run(n):
        t0 = time.perf_counter()
        for _ in range(n):
            T = np.eye(4)
            m.Match(cl, T, update_map=False)
        return (time.perf_counter() - t0) / n * 1e3
    run(5)
    ms = run(40)
    print(json.dumps({"kind": kind, "FLS_DEVICE_VOXELGRID": os.environ.get("FLS_DEVICE_VOXELGRID", "1"), "n_raw": int(cfg["scan"].shape[0]),
                      "n_filtered": int(m.stats.n_source), "iterations": int(m.stats.iterations), "ms_per_match_host_buffers": round(ms, 3),
                      "device_runs": m.map_size(105), "host_runs": m.map_size(106)}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for kind in ("icp", "ndt"):
            for dev in ("0", "1", "2"):  # host filter | device, std::sort order (default) | device, index order
                subprocess.run([sys.executable, __file__, kind], env=dict(os.environ, FLS_DEVICE_VOXELGRID=dev), check=False)
