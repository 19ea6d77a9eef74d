"""The brick pool of the iVox device image has a byte budget: a map whose bricks would cost more than FLS_IVOX_BRICK_BUDGET_MB (default
16 GB; a brick of the pool is ~26 KB with its AddPoints side arrays) takes the per-voxel hash-table image with host-maintained AddPoints instead
of failing a device allocation -- slower, same results.  The budget is read once per process, so the squeezed run is a process of its own."""
import json
import os
import subprocess
import sys

import pytest

from funny_lidar_slam_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["FLS_ROOT"])
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
cfg = synth.make_config(1, scale=0.1)
m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
m.AddCloudToLocalMap([cfg["map"]])
out = []
T = cfg["T_init"].copy()
for k in range(3):   # mapping mode: Match + the map update inside
    Tk = T.copy()
    ok = m.Match(reg.PointcloudCluster(planar_cloud_=cfg["scan"]), Tk, update_map=True)
    out.append([bool(ok), int(m.stats.iterations), int(m.stats.n_valid), Tk.tolist(), int(m.map_size(0))])
print("RESULT " + json.dumps({"runs": out, "over_budget": int(m.map_size(132)), "device_batches": int(m.map_size(103))}))
'''


def run(budget):
    env = dict(os.environ, FLS_ROOT=ROOT)
    if budget is not None:
        env["FLS_IVOX_BRICK_BUDGET_MB"] = str(budget)
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])


def test_map_beyond_the_brick_budget_takes_the_hash_table_image(built):
    assert _lib.device_count() >= 1
    normal, squeezed = run(None), run(1)
    assert normal["over_budget"] == 0 and squeezed["over_budget"] == 1
    assert squeezed["device_batches"] == 0  # (the hash-table image has no device AddPoints: the host maintains the map)
    assert normal["runs"] == squeezed["runs"], (normal["runs"], squeezed["runs"])  # return value, iterations, n_valid, pose bits, map size: identical
