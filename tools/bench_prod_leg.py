"""bench.py's `mapping_mode.filtered_planar_cloud_0p5m` leg on its own (the adapter's real iVox call, six scans), per-scan times printed: for A/B runs of
two library builds (FLS_REG_LIB) on one box.  usage: python tools/bench_prod_leg.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(1)
for rep in range(3):
    r = bench.bench_mapping_mode(reg, synth, cfg)
    f = r["filtered_planar_cloud_0p5m"]
    print(os.path.basename(os.environ.get("FLS_REG_LIB", "libfls_reg.so")), "rep", rep, "prod leg ms", round(f["ms_per_scan_match_plus_update_from_host_buffers"], 4), "iters", f["gn_iterations"],
          "| device_addpoints ms", round(r["device_addpoints"]["ms_per_scan_match_plus_update"], 4), flush=True)
