import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib
L = _lib.lib()
t0 = time.time()
print("devices", _lib.device_count(), "init %.1f s" % (time.time() - t0), flush=True)
kk = np.arange(8, dtype=np.uint32); vv = np.arange(8, dtype=np.uint32)
print("host sort rc", L.fls_debug_exact_sort(0, kk.ctypes.data_as(C.POINTER(C.c_uint32)), vv.ctypes.data_as(C.POINTER(C.c_uint32)), 8, 1), flush=True)
for n in [int(a) for a in sys.argv[1:]] or [2]:
    k = np.random.default_rng(1).integers(0, 5, n).astype(np.uint32); v = np.arange(n, dtype=np.uint32)
    res = []
    th = threading.Thread(target=lambda: res.append(L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), n, 0)), daemon=True)
    t0 = time.time()
    th.start()
    th.join(12.0)
    marks = (C.c_uint * 12)()
    rc = L.fls_debug_exact_sort_marks(marks, 12)
    print("n", n, "finished", not th.is_alive(), "in %.2f s" % (time.time() - t0), "rc", res, "marks rc", rc, list(marks), flush=True)
    if th.is_alive():
        break
os._exit(0)
