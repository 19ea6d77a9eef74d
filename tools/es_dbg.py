import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from funny_lidar_slam_amd import _lib
L = _lib.lib()
for n, hi in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(3000, 1 << 30)]:
    k = np.random.default_rng(n + 1).integers(0, hi, n).astype(np.uint32); v = np.arange(n, dtype=np.uint32)
    rc = L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), n, 0)
    print("n", n, "hi", hi, "rc", rc, "sorted", bool(np.all(np.diff(k.astype(np.int64)) >= 0)), flush=True)
