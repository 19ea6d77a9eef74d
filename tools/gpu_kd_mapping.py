"""mapping_mode.icp_optimized / .loam_full legs of bench.py on their own (Match + keyframe update from host buffers: host path,
default, opt-in device path); prints the JSON block."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from funny_lidar_slam_amd import registration as reg, synth
print(json.dumps(bench.bench_kd_mapping_mode(reg, synth), indent=1))
