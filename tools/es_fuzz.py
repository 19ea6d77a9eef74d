"""Fuzz of the device exact sort against libstdc++'s std::sort (same C entry point, on_host = 1): random sizes around every regime boundary
(16 / 64 / 2,048 / 4,096 / 32,768 / 131,072 records), key distributions from all-equal to all-distinct, and LiDAR-like piecewise-monotone
leaf indices (what drives introsort into its lopsided recursion and its heap-sort fallback).  usage: python tools/es_fuzz.py [n_cases] [seed] [device mode: 0 = host-steered sequence, 2 = the one-stream VoxelGrid's pre-enqueued sequence] [max n]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib

L = _lib.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
dev_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
max_n = int(sys.argv[4]) if len(sys.argv) > 4 else 140000
edges = [16, 17, 64, 65, 2047, 2048, 2049, 4096, 4097, 8192, 8193, 32768, 32769, 65536, 131072, 131073] + ([262144, 524288, 524289] if max_n > 600000 else [])
bad = declined = 0
t0 = time.time()
for case in range(n_cases):
    kind = case % 5
    n = int(rng.choice(edges)) + int(rng.integers(-3, 4)) if case % 3 == 0 else int(rng.integers(2, max_n))
    n = max(n, 1)
    if kind == 0:
        key = rng.integers(0, int(rng.choice([2, 5, 100, n // 3 + 2, 1 << 30])), n)
    elif kind == 1:  # ring-major LiDAR-like leaf indices: piecewise monotone with jitter
        rings = int(rng.integers(4, 64)); per = n // rings + 1
        az = np.tile(np.arange(per), rings)[:n]
        ring = np.repeat(np.arange(rings), per)[:n]
        key = (np.abs(np.sin(az * 2 * np.pi / per)) * 3000).astype(np.int64) + ring * int(rng.integers(1, 4000)) + rng.integers(0, 3, n)
    elif kind == 2:
        key = np.sort(rng.integers(0, n // 2 + 2, n)); key = key[::-1].copy() if case % 2 else key
    elif kind == 3:  # organ pipe / sawtooth
        key = np.minimum(np.arange(n), n - np.arange(n)) // int(rng.integers(1, 9)) if case % 2 else np.arange(n) % int(rng.integers(2, 5000))
    else:
        key = np.full(n, 7) if case % 10 == 4 else rng.integers(0, n + 1, n) // int(rng.integers(1, 50))
    key = np.asarray(key, np.int64) % (1 << 31)
    if os.environ.get("FLS_FUZZ_VERBOSE"):
        print("case", case, "kind", kind, "n", n, "keys", int(key.min()), int(key.max()), flush=True)
    res = []
    for on_host in (1, dev_mode):
        k, v = key.astype(np.uint32).copy(), np.arange(n, dtype=np.uint32)
        rc = L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), n, on_host)
        res.append((rc, k, v))
    (rh, kh, vh), (rd, kd, vd) = res
    if rd == _lib.FLS_ERR_STATE:
        declined += 1  # (a long range at introsort's depth limit: the callers take the host filter)
        continue
    if rh != 0 or rd != 0 or not np.array_equal(kh, kd) or not np.array_equal(vh, vd):
        bad += 1
        print("MISMATCH case", case, "kind", kind, "n", n, "rc", rh, rd, "first diff", int(np.flatnonzero(vh != vd)[0]) if rd == 0 and (vh != vd).any() else -1, flush=True)
print(f"exact-sort fuzz (device mode {dev_mode}): {n_cases} cases, {bad} mismatches, {declined} declined by the device, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
