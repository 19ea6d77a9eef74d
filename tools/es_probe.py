"""probe of the device exact sort, one size per process under a hard timeout (a hang must not eat the GPU budget)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "one":
    import ctypes as C, numpy as np, time
    from funny_lidar_slam_amd import _lib
    n, hi = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(n + 1)
    key = rng.integers(0, hi, n).astype(np.uint32)
    L = _lib.lib(); out = []
    for on_host in (1, 0):
        k, v = key.copy(), np.arange(n, dtype=np.uint32)
        t = time.perf_counter()
        rc = L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), n, on_host)
        out.append((rc, k, v, time.perf_counter() - t))
    print(n, hi, "rc", out[0][0], out[1][0], "equal", bool(np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][1], out[1][1])), "ms %.2f %.2f" % (1e3 * out[0][3], 1e3 * out[1][3]), flush=True)
else:
    for n, hi in [(2, 3), (17, 3), (100, 7), (5000, 50), (9000, 100), (20000, 300), (115200, 40000), (300001, 1000), (1400000, 200000)]:
        try:
            p = subprocess.run([sys.executable, __file__, "one", str(n), str(hi)], capture_output=True, text=True, timeout=25)
            print(p.stdout.strip() or p.stderr.strip()[-300:], flush=True)
        except subprocess.TimeoutExpired:
            print(n, hi, "HANG (25 s)", flush=True)
            break
