"""Phase stamps of the one-launch levels of the exact sort (kernels_exactsort_level.hpp): FLS_ES_DEBUG=1 python tools/es_level_stamps.py [n ...]
sorts the leaf indices of a ring-major scan (n = 115200) or of a concatenated deque (any other n: the scan tiled) through the test hook's mode 2 and
prints, per level, the 100 MHz stamps of the middle tile's thread 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib, synth
L = _lib.lib()
cfg = synth.make_config(2)
p = cfg["scan"].astype(np.float32)
q = np.floor(p * (np.float32(1.0) / np.float32(0.2))).astype(np.int64); q -= q.min(0); d = q.max(0) + 1
key0 = (q[:, 0] + q[:, 1] * d[0] + q[:, 2] * d[0] * d[1]).astype(np.uint32)
for n in [int(a) for a in sys.argv[1:]] or [115200]:
    key = np.tile(key0, n // key0.size + 1)[:n].copy()
    for rep in range(3):
        k, v = key.copy(), np.arange(n, dtype=np.uint32)
        rc = L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), n, 2)
        print("n", n, "rep", rep, "rc", rc, flush=True)
