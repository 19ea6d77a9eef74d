"""which of the bench's preprocessing filters does the exact device sort decline, and why (FLS_ES_DEBUG=1 prints the sort's state)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from funny_lidar_slam_amd import _lib, synth
cfg = synth.make_config(1)
rng = synth.rng_for(1, 123)
Tgt = cfg["T_gt"].copy()
clouds = [("map 1e6", cfg["map"])]
for k in range(6):
    clouds.append((f"scan {k}", synth.cast_scan(cfg["scene"], Tgt, rng=rng, **synth.VELODYNE_64)))
    Tgt = Tgt @ synth.random_pose(rng, 0.5, 0.5)
L = _lib.lib()
for name, c in clouds:
    for leaf in (0.5, 0.2):
        a = np.ascontiguousarray(c, np.float32)
        out = np.zeros((a.shape[0], 4), np.float32); n_out = C.c_size_t(0); fp = C.POINTER(C.c_float)
        rc = L.fls_voxel_grid_cloud(0, 1, a.ctypes.data_as(fp), a.shape[0], a.shape[1], np.float32(leaf), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
        print(name, "leaf", leaf, "n", a.shape[0], "rc", rc, "n_out", n_out.value, flush=True)
