"""std::sort's permutation on the device (csrc/kernels_exactsort.hpp) against std::sort itself (libstdc++, through the same C entry
point with on_host = 1): records {key, value} compared by key only, so the order of equal keys is introsort's business -- and what
pcl::VoxelGrid's leaf sums depend on.  The device result must equal the host's record for record."""
import ctypes as C

import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


# device modes of the test hook: 0 = the host-steered sequence (DeviceExactSort::run), 2 = what the one-stream VoxelGrid queues (fused_launch: the
# pre-enqueued levels above 32,768 records, then the task kernel -- no host round trip)
DEV_MODES = (0, 2)


def sort_both(key, dev_mode=0):
    L = _lib.lib()
    val = np.arange(key.size, dtype=np.uint32)
    out = []
    for on_host in (1, dev_mode):
        k, v = key.astype(np.uint32).copy(), val.copy()
        rc = L.fls_debug_exact_sort(0, k.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_uint32)), k.size, on_host)
        out.append((rc, k, v))
    return out


def check(key, label, modes=DEV_MODES):
    for dev_mode in modes:
        (rh, kh, vh), (rd, kd, vd) = sort_both(key, dev_mode)
        assert rh == 0 and rd == 0, (label, dev_mode, rh, rd)
        assert np.array_equal(kh, kd), (label, dev_mode)
        bad = np.flatnonzero(vh != vd)
        assert bad.size == 0, (label, dev_mode, key.size, bad[:5], vh[bad[:5]], vd[bad[:5]])


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 33, 100, 1000, 4095, 4096, 4097, 8193, 20000, 115200, 300001])
def test_random_keys_with_ties(n):
    rng = np.random.default_rng(n + 1)
    for hi in (3, max(2, n // 50 + 1), max(2, n // 3 + 1), 1 << 30):
        check(rng.integers(0, hi, n), f"n={n} range={hi}")


def test_structured_inputs():
    rng = np.random.default_rng(5)
    n = 60000
    base = rng.integers(0, 5000, n)
    check(np.sort(base), "sorted")
    check(np.sort(base)[::-1].copy(), "reversed")
    check(np.zeros(n, np.int64), "all equal")
    check(np.repeat(np.arange(n // 6), 6), "runs of six")
    check(np.tile(np.arange(300), n // 300), "sawtooth")
    k = np.sort(base); k[:: n // 7] += 3
    check(k, "nearly sorted")


def test_depth_limit_ranges_are_heap_sorted_like_libstdcxx():
    """A sorted array with every 97th key replaced at random drives introsort to its depth limit (2 log2 n partitions deep on a 67-record
    range: checked with the CPU model), where libstdc++ heap-sorts the range; ring-major scans at a 0.2 m leaf do the same on most frames.
    The device restates the heap routines (es_heap_sort): record for record the host's result."""
    rng = np.random.default_rng(5)
    n = 60000
    for step in (97, 1501):
        k = np.sort(rng.integers(0, 5000, n)); k[::step] = rng.integers(0, 5000, k[::step].size)
        check(k, f"nearly sorted, every {step}th key random")
    scene = synth.make_scene()
    srng = synth.rng_for(1, 123)
    Tgt = np.eye(4)
    for f in range(4):
        c = synth.cast_scan(scene, Tgt, rng=srng, **synth.VELODYNE_64).astype(np.float32)
        Tgt = Tgt @ synth.random_pose(srng, 0.5, 0.5)
        for leaf in (0.2, 0.5):
            q = np.floor(c * (np.float32(1.0) / np.float32(leaf))).astype(np.int64)
            q -= q.min(0)
            d = q.max(0) + 1
            check(q[:, 0] + q[:, 1] * d[0] + q[:, 2] * d[0] * d[1], f"frame {f} leaf {leaf}")


def test_leaf_indices_of_a_real_scan():
    """the keys the VoxelGrid actually sorts: leaf indices of a ring-major 64 x 1800 scan at the NDT / ICP leaf sizes"""
    cfg = synth.make_config(2)
    for leaf in (0.5, 0.1, 1.0):
        p = cfg["scan"].astype(np.float32)
        inv = np.float32(1.0) / np.float32(leaf)
        q = np.floor(p * inv).astype(np.int64)
        q -= q.min(0)
        d = q.max(0) + 1
        check(q[:, 0] + q[:, 1] * d[0] + q[:, 2] * d[0] * d[1], f"scan leaf={leaf}")


def test_large_cloud_many_levels():
    rng = np.random.default_rng(9)
    check(rng.integers(0, 200000, 1400000), "1.4 M records (a LoamFull planar deque)")


def test_pre_enqueued_sequence_large_clouds():
    """the keyframe deques' size class through the sequence the one-stream VoxelGrid queues (test hook mode 2): 600 k, 1.55 M and 2.2 M records"""
    rng = np.random.default_rng(77)
    check(rng.integers(0, 90000, 600000), "600 k records", modes=(2,))
    check(np.repeat(rng.integers(0, 40000, 100000), 16)[:1555200], "1.55 M records, runs of 16", modes=(2,))
    check(rng.integers(0, 300000, 2200000), "2.2 M records", modes=(2,))


def test_fuzz_against_std_sort():
    """tools/es_fuzz.py: 300 arrays around every regime boundary of the device sort (16 / 64 / 2,048 / 4,096 / 32,768 / 131,072 records), key
    distributions from all-equal to all-distinct, LiDAR-like piecewise-monotone keys, sorted / reversed / organ-pipe / sawtooth inputs: every
    sort the device accepts equals std::sort's record for record (adversarial inputs that hit introsort's depth limit on a long range are
    declined -- FLS_ERR_STATE -- and counted, not compared)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "es_fuzz.py"), "300", "4242"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    tail = run.stdout.strip().splitlines()[-1]
    assert " 0 mismatches" in tail, tail
    print(tail)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "es_fuzz.py"), "200", "777", "2", "700000"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    tail = run.stdout.strip().splitlines()[-1]
    assert " 0 mismatches" in tail, tail
    print(tail)
