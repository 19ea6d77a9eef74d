"""Device VoxelGrid (csrc/kernels_voxelgrid.hpp + kernels_exactsort.hpp) against the oracle's pcl::VoxelGrid
restatement (oracle/flo_cloud.h, itself pinned by the compiled reference's VoxelGridCloud in tests/test_ref_pin.py).

Round 4: the device filter is the DEFAULT and BIT-IDENTICAL to the reference -- its sort reproduces std::sort's permutation
(tests/test_gpu_exact_sort.py), so every leaf is summed in the reference's order.  `check` asserts bit equality of the whole output.
FLS_DEVICE_VOXELGRID=2 keeps the round-2/3 form (stable radix sort, leaf sums in ascending point index) for A/B; its contract --
leaves / order / integers exact, leaves of <= 2 points bit-identical, larger leaves within count * 2^-23 * max|coordinate| -- is what
`check_index_order` asserts.  FLS_DEVICE_VOXELGRID=0 = the exact host filter (worker pool).
"""
import ctypes as C

import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


def device_voxel_grid(cloud, leaf, stride=None):
    a = np.ascontiguousarray(cloud, np.float32)
    n, s = a.shape
    out = np.zeros((max(n, 1), 4), np.float32)
    n_out = C.c_size_t(0)
    fp = C.POINTER(C.c_float)
    rc = _lib.lib().fls_debug_voxel_grid(0, a.ctypes.data_as(fp), n, s, np.float32(leaf), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
    return rc, out[: n_out.value].copy()


def leaf_counts(cloud, leaf):
    """points per output leaf, in output order (numpy restatement of the integer part of the filter)"""
    a = np.asarray(cloud, np.float32)
    fin = np.isfinite(a[:, :3]).all(1)
    p = a[fin, :3]
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = p.min(0), p.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    div_b = np.floor(mx * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(p * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    u, inv_idx, cnt = np.unique(idx, return_inverse=True, return_counts=True)
    amax = np.zeros(len(u), np.float32)
    np.maximum.at(amax, inv_idx, np.abs(a[fin]).max(1))
    return cnt, amax


def check(cloud, leaf):
    """default mode: the device output IS the reference's, bit for bit"""
    ref = O.voxel_grid(cloud, leaf)
    rc, out = device_voxel_grid(cloud, leaf)
    assert rc == 0, rc
    assert out.shape == ref.shape, (out.shape, ref.shape)
    cnt, amax = leaf_counts(cloud, leaf)
    assert len(cnt) == ref.shape[0]
    same = (out.view(np.uint32) == ref.view(np.uint32)).all(1)
    assert same.all(), (int((~same).sum()), int(len(same)), np.flatnonzero(~same)[:5], cnt[~same][:5])
    return dict(leaves=int(len(cnt)), big=int((cnt > 2).sum()), max_leaf=int(cnt.max()), identical=1.0, max_ulp=0.0)


def check_index_order(cloud, leaf):
    """FLS_DEVICE_VOXELGRID=2 (set by the caller): the round-2/3 contract"""
    ref = O.voxel_grid(cloud, leaf)
    rc, out = device_voxel_grid(cloud, leaf)
    assert rc == 0, rc
    assert out.shape == ref.shape, (out.shape, ref.shape)
    cnt, amax = leaf_counts(cloud, leaf)
    assert len(cnt) == ref.shape[0]
    small = cnt <= 2
    assert np.array_equal(out[small].view(np.uint32), ref[small].view(np.uint32)), "leaves with <= 2 points must be bit-identical"
    bound = (cnt.astype(np.float64) * 2.0 ** -23 * amax)[:, None]
    d = np.abs(out.astype(np.float64) - ref.astype(np.float64))
    assert (d <= bound).all(), float((d / np.maximum(bound, 1e-30)).max())
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return dict(leaves=int(len(cnt)), big=int((~small).sum()), identical=float((out.view(np.uint32) == ref.view(np.uint32)).all(1).mean()),
                max_ulp=float((d / ulp).max()))


def test_index_order_mode_keeps_its_contract(monkeypatch):
    monkeypatch.setenv("FLS_DEVICE_VOXELGRID", "2")
    cfg = synth.make_config(2, with_map=False)
    scan = np.concatenate([cfg["scan"][:, :3], np.linspace(0, 1, cfg["scan"].shape[0], dtype=np.float32)[:, None]], axis=1)
    r = check_index_order(scan, 0.2)
    print("index-order mode:", r)
    assert r["leaves"] > 1000 and r["max_ulp"] <= 16 and r["identical"] < 1.0  # (it really is the other summation order)


@pytest.mark.parametrize("cid,leaf", [(0, 0.4), (2, 0.2), (1, 0.5)])
def test_scan_filters_of_the_benchmark_configs(cid, leaf):
    """the source filters Match runs: configs[0] (16x900 scan, 0.4 m), configs[2] (64x1800 scan, 0.2 m), and a coarse one"""
    cfg = synth.make_config(cid, with_map=False)
    scan = np.concatenate([cfg["scan"][:, :3], np.linspace(0, 1, cfg["scan"].shape[0], dtype=np.float32)[:, None]], axis=1)
    r = check(scan, leaf)
    print(cid, leaf, r)
    assert r["leaves"] > 1000 and r["big"] > 100


def test_random_clouds_edge_cases():
    rng = np.random.default_rng(77)
    checked = declined = 0
    for case in range(16):
        n = int(rng.integers(1, 40000))
        a = (rng.normal(size=(n, 4)) * rng.choice([0.5, 5.0, 60.0])).astype(np.float32)
        if case % 3 == 0:  # heavy leaves: many points in few cells
            a[:, :3] = (rng.integers(-3, 3, size=(n, 3)) + rng.random((n, 3)) * 0.999).astype(np.float32)
        if case % 4 == 1:  # non-finite points are dropped (voxel_grid.hpp:98-101)
            bad = rng.integers(0, n, size=max(1, n // 50))
            a[bad, rng.integers(0, 3, size=bad.size)] = rng.choice([np.nan, np.inf, -np.inf], size=bad.size)
        if case == 5:
            a = a[:1]
        if case == 6:      # exact cell boundaries and negative zero
            a[:, :3] = np.round(a[:, :3] * 2) / 2
            a[::7, 0] = -0.0
        leaf = float(rng.choice([0.1, 0.25, 0.5, 1.0, 3.0]))
        fin = a[np.isfinite(a[:, :3]).all(1), :3]
        if len(fin) == 0:
            continue
        box = np.prod(np.floor((fin.max(0) - fin.min(0)).astype(np.float64) / leaf) + 1)
        if box > 2 ** 31 - 1:  # "leaf size too small": declined (the host filter copies the input)
            assert device_voxel_grid(a, leaf)[0] == _lib.FLS_ERR_STATE
            declined += 1
            continue
        if len(fin) != len(a):  # the reference sorts the finite points only: a cloud with non-finite points is the host filter's
            assert device_voxel_grid(a, leaf)[0] == _lib.FLS_ERR_STATE
            declined += 1
            continue
        r = check(a, leaf)
        assert r["leaves"] >= 1
        checked += 1
    assert checked >= 8 and declined >= 1, (checked, declined)


@pytest.mark.parametrize("copies,leaf", [(3, 0.4), (5, 1.0), (13, 0.4)])
def test_keyframe_deque_clouds_dense_leaves_bit_identical(copies, leaf):
    """The map-side filter of the kd-tree kinds (icp_optimized.h:187, loam_full_kdtree.h:92-100): the concatenated keyframe deque -- clouds beyond 131,072
    points (the host-steered levels of the exact sort, 8,192-record LDS ranges, round 6), the same surfaces seen from neighbouring poses, leaves of
    hundreds to thousands of points (summed by the head's wave: vg_centroid_body's long-run path, incl. partial last rounds).  Every leaf bit-identical."""
    cfg = synth.make_config(2, with_map=False)
    base = cfg["scan"][::2, :3] if copies > 5 else cfg["scan"][:, :3]
    rng = np.random.default_rng(1000 + copies)
    parts = []
    for k in range(copies):  # a keyframe every ~1 m: the same cloud shifted by a few centimetres to a metre (k = 0, 1: exact duplicates)
        shift = np.float32(0.0) if k < 2 else rng.uniform(-1.0, 1.0, size=3).astype(np.float32)
        parts.append((base + shift).astype(np.float32))
    cloud = np.concatenate(parts, axis=0)
    cloud = np.concatenate([cloud, rng.random((cloud.shape[0], 1), dtype=np.float32)], axis=1)
    assert cloud.shape[0] > 131072
    r = check(cloud, leaf)
    print(copies, leaf, cloud.shape[0], r)
    assert r["max_leaf"] >= 128 and r["big"] > 1000


def test_sparse_cloud_is_bit_identical():
    """every leaf holds at most two points: the whole output equals the reference's bit for bit"""
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    a = g + rng.random(g.shape, dtype=np.float32) * 0.9
    b = a[rng.permutation(len(a))[: len(a) // 2]] + np.float32(0.01)  # a second point in half of the cells
    b = b[(np.floor(b) == np.floor(b - np.float32(0.01))).all(1)]
    cloud = np.concatenate([a, b])[rng.permutation(len(a) + len(b))]
    cloud = np.concatenate([cloud, rng.random((len(cloud), 1), dtype=np.float32)], axis=1)
    ref = O.voxel_grid(cloud, 1.0)
    rc, out = device_voxel_grid(cloud, 1.0)
    assert rc == 0 and np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert leaf_counts(cloud, 1.0)[0].max() == 2


def test_declined_inputs():
    """no finite point, and PCL's "leaf size too small" case (index would overflow int): the device path declines, the matchers
    take the host filter (which copies the input in the second case)"""
    a = np.full((10, 4), np.nan, np.float32)
    assert device_voxel_grid(a, 0.5)[0] == _lib.FLS_ERR_STATE
    b = np.array([[0, 0, 0, 0], [1e6, 1e6, 1e6, 0]], np.float32)
    assert device_voxel_grid(b, 0.1)[0] == _lib.FLS_ERR_STATE


@pytest.mark.parametrize("mode,y,cid,loc", [("IcpOptimized", reg.YAML_NCLT_ICP, 0, True), ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False)])
def test_match_with_device_source_filter(mode, y, cid, loc, monkeypatch):
    """ICP / NDT Match with the device source filter (default) against the host filter (FLS_DEVICE_VOXELGRID=0): the filtered clouds
    are bit-identical, so the two handles return IDENTICAL poses; then a map update from the device-resident filtered cloud
    (mapping-mode NDT)."""
    cfg = synth.make_config(cid, scale=1.0 if cid == 0 else 0.2)
    res = {}
    for dev in (0, 1):
        monkeypatch.setenv("FLS_DEVICE_VOXELGRID", str(dev))
        m = reg.make_matcher(mode, y, is_localization_mode=loc)
        m.AddCloudToLocalMap([cfg["map"]])
        T = np.eye(4)
        ok = m.Match(util.cluster_for(mode, cfg["scan"]), T, update_map=not loc)
        T2 = T.copy()
        ok2 = m.Match(util.cluster_for(mode, cfg["scan"]), T2, update_map=not loc)
        res[dev] = (ok, T.copy(), m.stats.n_source, m.stats.iterations, ok2, T2.copy(), m.map_size(0), m.map_size(105), m.map_size(106),
                    m.map_size(111) if mode == "IncrementalNDT" else 0)
        m.close()
    h, d = res[0], res[1]
    assert h[7] == 0 and h[8] == 2 and d[7] == 2 and d[8] == 0  # which filter ran
    assert h[0] == d[0] and h[4] == d[4] and h[2] == d[2] and h[3] == d[3]
    for a, b in ((h[1], d[1]), (h[5], d[5])):
        assert np.array_equal(a, b), synth.pose_error(a, b)
    assert h[6] == d[6]  # map size after the updates
    if mode == "IncrementalNDT":  # with the device filter the whole update chain (transform, second VoxelGrid, UpdateVoxel) stays on the device
        assert h[9] == 0 and d[9] == 2, (h[9], d[9])


def test_voxel_grid_cloud_entry_point_device_mode():
    """fls_voxel_grid_cloud(FLS_VOXELGRID_DEVICE) = the preprocessing / loop-closure filter entry point on the GPU: same output as the
    test hook, the contract against the exact mode, and the documented fall-back when the device path declines."""
    cfg = synth.make_config(1, scale=0.3)
    cloud = np.concatenate([cfg["scan"], np.linspace(0, 1, len(cfg["scan"]), dtype=np.float32)[:, None]], axis=1)
    dev = reg.VoxelGridCloud(cloud, 0.5, on_device=True)
    exact = reg.VoxelGridCloud(cloud, 0.5, on_device=False)
    rc, hook = device_voxel_grid(cloud, 0.5)
    assert rc == 0 and np.array_equal(dev.view(np.uint32), hook.view(np.uint32))
    assert dev.shape == exact.shape and np.array_equal(dev.view(np.uint32), exact.view(np.uint32))
    far = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2]], np.float32)  # "leaf size too small": the device declines, the exact filter copies the input
    assert np.array_equal(reg.VoxelGridCloud(far, 0.1, on_device=True), far)


@pytest.mark.parametrize("mode,y,cid,loc", [("IcpOptimized", reg.YAML_NCLT_ICP, 0, True), ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False)])
def test_scan_upload_raw_filters_inside_every_resident_match(mode, y, cid, loc):
    """fls_scan_upload_raw (round 5): the raw cloud is resident and EVERY fls_match_resident runs the source VoxelGrid first, like the
    reference's Match (icp_optimized.h:57, incremental_ndt.h:231-232) -- identical to fls_match from host buffers, call after call,
    and to the filtered-at-upload form.  This is the call bench.py times for configs[0] / [2]."""
    cfg = synth.make_config(cid, scale=1.0 if cid == 0 else 0.2)
    cl = util.cluster_for(mode, cfg["scan"])
    a = reg.make_matcher(mode, y, is_localization_mode=loc)
    b = reg.make_matcher(mode, y, is_localization_mode=loc)
    for m in (a, b):
        m.AddCloudToLocalMap([cfg["map"]])
    b.UploadScanRaw(cl)
    for call in range(3):
        Ta, Tb = np.eye(4), np.eye(4)
        oka = a.Match(cl, Ta, update_map=False)
        okb = b.MatchResident(Tb, update_map=False)
        assert oka == okb and a.stats.iterations == b.stats.iterations and a.stats.n_source == b.stats.n_source
        assert np.array_equal(Ta, Tb), (call, synth.pose_error(Ta, Tb))
    assert b.map_size(105) == 3 and b.map_size(106) == 0  # three device filters, none on the host
    a.UploadScan(cl)
    Tc = np.eye(4)
    a.MatchResident(Tc, update_map=False)
    assert np.array_equal(Tc, Tb)
    a.close(); b.close()


def test_fused_launch_sequence_equals_the_round4_sequence(monkeypatch):
    """FLS_VG_FUSED=0 restores the round-4 host-steered sequence (bounds read back, queue initialised by the runtime, two stream
    synchronisations); the default derives the plan on the device and publishes the verdict to a mailbox.  Same bits either way,
    including what both decline."""
    cfg = synth.make_config(2, scale=0.3)
    cloud = np.concatenate([cfg["scan"], np.linspace(0, 1, len(cfg["scan"]), dtype=np.float32)[:, None]], axis=1)
    bad = cloud.copy(); bad[5, 1] = np.nan
    far = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2]], np.float32)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("FLS_VG_FUSED", fused)
        res[fused] = [device_voxel_grid(c, leaf) for c, leaf in ((cloud, 0.2), (cloud[:1], 0.2), (cloud[:700], 1.0), (bad, 0.2), (far, 0.1))]
    for (rc1, o1), (rc0, o0) in zip(res["1"], res["0"]):
        assert rc1 == rc0 and np.array_equal(o1.view(np.uint32), o0.view(np.uint32))
    assert [r[0] for r in res["1"]] == [0, 0, 0, _lib.FLS_ERR_STATE, _lib.FLS_ERR_STATE]
    ref = O.voxel_grid(cloud, 0.2)
    assert np.array_equal(res["1"][0][1].view(np.uint32), ref.view(np.uint32))
