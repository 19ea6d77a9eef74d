"""GPU parity tests (the tests proper): HIP path through the C ABI vs the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): correspondence indices bit-exact, SE(3) pose within 1e-4 m / 1e-4 rad
after the same iteration count.  The tests are stricter: n_valid and the pose are compared after EVERY
Gauss-Newton iteration, the valid flags and neighbour counts of every source point must be identical.
"""
import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


def run_pair(mode, y, map_clouds, scan, corner=None, loc=False, T_init=None, update_map=False, tie_ok=True, sets_only_tail=False):
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    o = util.oracle_for(mode, y, loc)
    m.AddCloudToLocalMap(map_clouds)
    o.AddCloudToLocalMap(*map_clouds)
    T = (np.eye(4) if T_init is None else T_init).copy()
    ok = m.Match(util.cluster_for(mode, scan, corner), T, update_map=update_map)
    ok_ref, T_ref = o.Match(scan, np.eye(4) if T_init is None else T_init, src1=corner, update_map=update_map)
    ties = int(o.counters().tie_queries) if tie_ok else 0
    slots = (0, 1) if mode == "LoamFull_KdTree" else (0,)
    dt, dr = util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, slots=slots, sets_only_tail=sets_only_tail, max_tie_rows=ties)
    return m, o, T, T_ref


@pytest.mark.parametrize("scale", [0.05, 1.0])
def test_config2_p2plane_ivox(scale):
    """BASELINE configs[1]: 64x1800 scan, point-to-plane into the 1e6-pt iVox map (also a 5% slice)."""
    cfg = synth.make_config(1, scale=scale)
    m, o, T, T_ref = run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [cfg["map"]], cfg["scan"], sets_only_tail=True)
    assert m.stats.converged == 1
    dt, dr = synth.pose_error(T, cfg["T_gt"])
    assert dt < 0.05 and dr < 0.005  # it actually registers (not only agrees with the oracle)


def test_device_traffic_counters_equal_the_oracles():
    """The roofline inputs (probes, hit voxels, candidate points of every iteration) counted by the kNN kernel's
    counting variant equal the oracle's instrumentation exactly -- on a fresh handle, and again on the steady state bench.py times
    (the same scan re-registered on one handle: nearest_points_ persists, Q15, so later calls differ from the first one)."""
    cfg = synth.make_config(1, scale=0.1)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([cfg["map"]])
    o.AddCloudToLocalMap(cfg["map"])
    m.set_profiling(False, counters=True)
    cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
    for call in range(3):
        T = np.eye(4)
        ok = m.Match(cl, T, update_map=False)
        ok_ref, T_ref = o.Match(cfg["scan"], np.eye(4), update_map=False)
        c = o.counters()
        assert m.traffic_counters() == (c.probes, c.hit_voxels, c.cand_points), (call, m.traffic_counters(), (c.probes, c.hit_voxels, c.cand_points))
        assert c.point_iters == cfg["scan"].shape[0] * o.stats.iterations and m.stats.iterations == o.stats.iterations
        good, dt, dr = util.pose_close(T, T_ref)
        assert good and ok == ok_ref, (call, dt, dr)
    m.close()


def test_repeated_match_on_one_handle_call_k_equals_oracle_call_k_full_size():
    """The computation bench.py TIMES -- the same resident 115,200-point scan re-registered on ONE handle, full BASELINE configs[1]
    size -- against the oracle driven the same way, call k vs call k.  nearest_points_ persists from Match to
    Match (Q15: a point without candidates keeps its previous list), so call k depends on every earlier call.  Per call: same return, iterations, n_valid, the three traffic counters EXACTLY, pose <= 1e-12 m / rad."""
    cfg = synth.make_config(1)
    y = reg.YAML_NCLT_IVOX
    m = reg.make_matcher("PointToPlane_IVOX", y)
    o = util.oracle_for("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([cfg["map"]])
    o.AddCloudToLocalMap(cfg["map"])
    cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
    m.UploadScan(cl)
    m.set_profiling(False, counters=True)  # the counting variant of the kNN kernel: same results, counters on the device
    seen = []
    for call in range(12):
        T = cfg["T_init"].copy()
        ok = m.MatchResident(T, update_map=False)
        ok_ref, T_ref = o.Match(cfg["scan"], cfg["T_init"], update_map=False)
        c = o.counters()
        got, want = m.traffic_counters(), (c.probes, c.hit_voxels, c.cand_points)
        dt, dr = synth.pose_error(T, T_ref)
        assert ok == ok_ref and m.stats.iterations == o.stats.iterations and m.stats.n_valid == o.stats.n_valid, (call, m.stats.iterations, o.stats.iterations, m.stats.n_valid, o.stats.n_valid)
        assert got == want, (call, got, want)
        assert dt <= 1e-12 and dr <= 1e-12, (call, dt, dr)
        seen.append((want, m.stats.n_valid))
    # the steady state is NOT a fixed point: the persisting neighbour lists make the repeated Match a period-2 cycle (two poses
    # 1.7e-5 m apart on this workload) -- round 2's bench compared the 27th call with the oracle's 3rd and read that as a gap.
    # From some call on, call k equals call k + 2.
    settled = next(k for k in range(len(seen) - 2) if all(seen[j] == seen[j + 2] for j in range(k, len(seen) - 2)))
    print("call-by-call (counters, n_valid):", seen[:6], "... period <= 2 from call", settled)
    assert settled <= 6, seen
    m.close()
    o.close()


def test_config1_icp_localization():
    """BASELINE configs[0]: 16x900 scan, Optimized-ICP vs the 50k-pt fixed map (localization mode), + GetFitnessScore."""
    cfg = synth.make_config(0)
    m, o, T, T_ref = run_pair("IcpOptimized", reg.YAML_NCLT_ICP, [cfg["map"]], cfg["scan"], loc=True)
    f, f_ref = m.GetFitnessScore(2.0), o.GetFitnessScore(2.0)
    assert f == pytest.approx(f_ref, rel=1e-6)


@pytest.mark.parametrize("scale", [0.1, 1.0])
def test_config3_incremental_ndt(scale):
    """BASELINE configs[2]: Incremental-NDT, 1.0 m voxels."""
    cfg = synth.make_config(2, scale=scale)
    run_pair("IncrementalNDT", reg.YAML_NCLT_NDT, [cfg["map"]], cfg["scan"])


@pytest.mark.parametrize("scale", [0.1, 1.0])
def test_config4_loam_full(scale):
    """BASELINE configs[3]: LOAM frontend, point-to-line (corner) + point-to-plane (surf) residuals."""
    cfg = synth.make_config(3, scale=scale)
    m, o, T, T_ref = run_pair("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, [cfg["map"], cfg["corner_map"]], cfg["scan"], corner=cfg["corner_scan"])
    assert m.stats.n_valid_corner > 0


@pytest.mark.parametrize("scale", [0.1, 1.0])
def test_p2plane_kdtree_localization(scale):
    """LoamPointToPlaneKdtree (localization only in the reference): un-gated exact 5-NN + fitness; scale 1.0 = the full 115,200-point
    scan against the 1e6-point map (round 3: the ring search beyond the 27 cells is walked by the whole 8-lane group)."""
    cfg = synth.make_config(1, scale=scale)
    m, o, T, T_ref = run_pair("PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, [cfg["map"]], cfg["scan"], loc=True)
    assert m.GetFitnessScore(2.0) == pytest.approx(o.GetFitnessScore(2.0), rel=1e-6)


def test_p2plane_ivox_localization_fitness():
    cfg = synth.make_config(1, scale=0.05)
    m, o, T, T_ref = run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [cfg["map"]], cfg["scan"], loc=True, sets_only_tail=True)
    assert m.GetFitnessScore(2.0) == pytest.approx(o.GetFitnessScore(2.0), rel=1e-6)
    m2 = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    assert m2.GetFitnessScore(2.0) == reg.FloatNaN  # mapping mode: FloatNaN (loam_point_to_plane_ivox.h:226-228)


def test_determinism_two_runs_bit_identical():
    cfg = synth.make_config(1, scale=0.1)
    outs = []
    for _ in range(2):
        m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        m.AddCloudToLocalMap([cfg["map"]])
        T = np.eye(4)
        m.Match(reg.PointcloudCluster(planar_cloud_=cfg["scan"]), T, update_map=False)
        outs.append((T.copy(), m.iteration_log()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1][0], outs[1][1][0]) and np.array_equal(outs[0][1][2], outs[1][1][2])


def test_pcl_layout_stride8_equals_packed():
    """pcl::PointXYZI memory (8 floats / point) through the ABI gives the same result as packed xyz."""
    cfg = synth.make_config(1, scale=0.05)

    def pad8(c):
        o = np.zeros((c.shape[0], 8), np.float32)
        o[:, :3] = c
        o[:, 3] = 1.0
        return o

    res = []
    for f in (lambda c: c, pad8):
        m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        m.AddCloudToLocalMap([f(cfg["map"])])
        T = np.eye(4)
        m.Match(reg.PointcloudCluster(planar_cloud_=f(cfg["scan"])), T, update_map=False)
        res.append(T.copy())
    assert np.array_equal(res[0], res[1])


def _replay(n_scans, capacity=None, monkeypatch=None, capacity_over_initial=None):
    """Mapping-mode replay along a short trajectory: Match -> AddCloudToLocalMap rule -> next Match."""
    scene = synth.make_scene()
    rng = synth.rng_for(1, 11)
    radius = 30.0
    mp = synth.sample_map(scene, 60000, synth.rng_for(1, 0, 5), radius=radius)
    lid = dict(synth.VELODYNE_64, n_az=60)
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    if capacity_over_initial is not None:  # LRU capacity = voxels of the initial map + this many (the reference hard-codes 1e6)
        probe = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        probe.AddCloudToLocalMap(mp)
        capacity = probe.map_voxels() + capacity_over_initial
        probe.close()
    if capacity is not None:
        monkeypatch.setenv("FLS_IVOX_CAPACITY", str(capacity))
        o.set_ivox_capacity(capacity)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([mp])
    o.AddCloudToLocalMap(mp)
    assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels()
    Tgt = np.eye(4)
    guess = np.eye(4)
    for k in range(n_scans):
        Tgt = Tgt @ synth.random_pose(rng, 1.0, 0.6)
        scan = synth.cast_scan(scene, Tgt, rng=rng, max_range=radius + 8.0, **lid)  # sees beyond the mapped disc: the map grows
        T = guess.copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=scan), T, update_map=True)
        ok_ref, T_ref = o.Match(scan, guess, update_map=True)
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        assert m.map_size() == o.map_size(), (k, m.map_size(), o.map_size())
        assert m.map_size(102) == o.map_voxels()
        guess = T_ref
    return m, o


def test_mapping_replay_incremental_image_updates():
    """8 scans with map growth: IVoxMap::AddPoints runs on the device (kernels_ivox_update.hpp: no per-scan read-back of
    decisions, no host insert, no re-flatten after the first build) and every Match still agrees with the oracle bit for bit
    in its correspondences -- ids included, i.e. the device assigns the reference's insertion order."""
    m, o = _replay(8)
    assert m.map_size(103) >= 7, "every map update should run on the device"
    assert m.map_size(104) == 0, "no batch refused by the device"
    assert m.map_size(101) == 1, "only the initial build is a full flatten"
    assert m.map_size() > 60000


def test_mapping_replay_host_path_ab(monkeypatch):
    """The exact sequential host path (FLS_IVOX_DEVICE_UPDATE=0, the round-1 path and the fallback of the device path)."""
    monkeypatch.setenv("FLS_IVOX_DEVICE_UPDATE", "0")
    m, o = _replay(4)
    assert m.map_size(103) == 0 and m.map_size(100) >= 3


def test_mapping_replay_device_refuses_batch_that_would_evict(monkeypatch):
    """LRU capacity a little above the initial map: the first scans are applied by the device; the batch whose voxel creations
    would reach the capacity is refused BEFORE any state changes, the device image (points, voxel table, LRU stamps) is read back
    into the host mirror and the exact sequential path -- with evictions -- takes over.  Parity with the oracle throughout."""
    monkeypatch.setenv("FLS_IVOX_DEVICE_MARGIN", "64")  # (head-room below the capacity needed to run on the device; default 4096)
    monkeypatch.setenv("FLS_IVOX_DEVICE_EVICT", "0")    # round-2 behaviour: evictions only on the host (the next tests run them on the device)
    m, o = _replay(8, monkeypatch=monkeypatch, capacity_over_initial=64 + 250)  # the scenario creates 86, 145, 190, ... 473 voxels
    assert m.map_size(103) >= 1, "some batches must have run on the device"
    assert m.map_size(104) >= 1, "a batch must have been refused and replayed on the host"
    assert m.map_size(102) == o.map_voxels()


def test_mapping_replay_with_lru_eviction(monkeypatch):
    """Small iVox capacity (test hook; the reference hard-codes 1e6 voxels): the LRU rule of ivox_map.cpp:133-136
    evicts voxels during the replay; evicted cells must disappear from the device image."""
    m, o = _replay(6, capacity=9000, monkeypatch=monkeypatch)
    assert o.map_voxels() <= 9000


def test_mapping_replay_device_evictions_straight_run(monkeypatch):
    """Round 3: LRU evictions INSIDE a device batch (ivox_map.cpp:133-136).  The map starts as one dense scan and grows along a
    straight 33 m run with a 20 m sensor range; the capacity (5,000 voxels, test hook) is reached after ~15 scans, from then on
    every scan creates 100-250 voxels and the same number of least recently touched ones are evicted ON THE DEVICE (alive cells
    listed and sorted by their 64-bit LRU stamp; a candidate the batch touches BEFORE its turn is skipped like the reference does,
    one it touches after its turn is evicted and re-created from the batch's points, like the reference does).  Every Match equals the oracle: ids, flags, n_valid, poses, map points and voxel counts."""
    from tests import replay
    cap = 5000
    monkeypatch.setenv("FLS_IVOX_CAPACITY", str(cap))
    start = np.eye(4)
    start[1, 3] = 18.0
    r = replay.make_replay("ivox", n_frames=40, yaw_long_deg=0.0, start=start, max_range=20.0)
    scene = synth.make_scene()
    s0 = synth.cast_scan(scene, start, rng=synth.rng_for(5, 99), max_range=20.0, **dict(synth.VELODYNE_64, n_az=600))
    init = (s0.astype(np.float64) @ start[:3, :3].T + start[:3, 3]).astype(np.float32)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o.set_ivox_capacity(cap)
    m.AddCloudToLocalMap([init])
    o.AddCloudToLocalMap(init)
    Tp = start.copy()
    at_cap = 0
    for k, f in enumerate(r["frames"]):
        guess = Tp @ f["guess_step"]
        T = guess.copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=f["scan"]), T, update_map=True)
        ok_ref, T_ref = o.Match(f["scan"], guess, update_map=True)
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels(), (k, m.map_size(), o.map_size(), m.map_size(102), o.map_voxels())
        at_cap += int(o.map_voxels() == cap - 1)
        Tp = T_ref
    applied, refused, evicted = m.map_size(103), m.map_size(104), m.map_size(117)
    print(f"ivox straight run at capacity {cap}: {applied} device batches, {refused} refused (eviction-order conflicts {m.map_size(119)}, point array full {m.map_size(120)}, "
          f"outside the window {m.map_size(121)}), {evicted} voxels evicted on the device, {at_cap} scans at the capacity")
    assert at_cap >= 15 and evicted > 1000, (at_cap, evicted)
    # a candidate that the batch touches AFTER its eviction is evicted and re-created by the reference (one more creation, one more
    # eviction, ...): such voxels exist in this scenario (old near-ground voxels still in range).  Round 3 refused those batches and
    # replayed them on the host; since round 4 the selection resolves them on the device (ivox_evict_select: the re-created voxel's first
    # rank joins the creation sequence; tests/host/evict_conflict_model_test.cpp) -- no batch leaves the device for an eviction-order conflict
    recreated = m.map_size(126)
    print(f"voxels evicted and re-created inside one device batch: {recreated}")
    assert m.map_size(119) == 0 and recreated >= 1, (m.map_size(119), recreated)
    assert applied >= 28 and refused == m.map_size(120) + m.map_size(121), (applied, refused)
    m.close()
    o.close()


def test_map_export_import_gives_an_identical_handle():
    """fls_map_export / fls_map_import (rank 0 builds the map, the image is broadcast): after a few mapping-mode
    scans handle A exports its map (which lives on the device by then), a fresh handle B imports the blob, and both are then driven
    through the same further scans: identical poses, ids and map sizes -- B is A, including the LRU order and the insertion ids."""
    scene = synth.make_scene()
    rng = synth.rng_for(1, 31)
    mp = synth.sample_map(scene, 50000, synth.rng_for(1, 0, 6), radius=28.0)
    lid = dict(synth.VELODYNE_64, n_az=50)
    y = reg.YAML_NCLT_IVOX
    a = reg.make_matcher("PointToPlane_IVOX", y)
    a.AddCloudToLocalMap([mp])
    Tgt, guess, scans = np.eye(4), np.eye(4), []
    for k in range(6):
        Tgt = Tgt @ synth.random_pose(rng, 1.0, 0.5)
        scans.append(synth.cast_scan(scene, Tgt, rng=rng, max_range=34.0, **lid))
    for k in range(3):
        T = guess.copy()
        a.Match(reg.PointcloudCluster(planar_cloud_=scans[k]), T, update_map=True)
        guess = T
    blob = a.ExportMap()
    assert blob.size > 16 * a.map_size()
    b = reg.make_matcher("PointToPlane_IVOX", y)
    b.ImportMap(blob)
    assert b.map_size() == a.map_size() and b.map_size(102) == a.map_size(102)
    # (1) the imported map IS the exporter's map: jobs with fresh per-job state (fls_match_batch) against both give identical results
    #     (A's own next Match would not: its nearest_points_ of the last scan persist -- Q15 -- and B has none)
    probe = [reg.PointcloudCluster(planar_cloud_=scans[k]) for k in (3, 4)]
    oka, Ta, sa = a.MatchBatch(probe, [guess, guess], lanes=2)
    okb, Tb, sb = b.MatchBatch(probe, [guess, guess], lanes=2)
    assert oka == okb and np.array_equal(Ta, Tb)
    assert [s.n_valid for s in sa] == [s.n_valid for s in sb] and [s.iterations for s in sa] == [s.iterations for s in sb]
    # (2) and it carries everything the map update needs (insertion ids, LRU order, counters): A re-imports its own blob (which clears
    #     its nearest_points_, like B's), then both run the same further mapping-mode scans -- bit-identical poses, ids, map sizes
    a.ImportMap(blob)
    for k in range(3, 6):
        Ta, Tb = guess.copy(), guess.copy()
        oka = a.Match(reg.PointcloudCluster(planar_cloud_=scans[k]), Ta, update_map=True)
        okb = b.Match(reg.PointcloudCluster(planar_cloud_=scans[k]), Tb, update_map=True)
        assert oka == okb and np.array_equal(Ta, Tb), k
        ia, ca, va = a.correspondences()
        ib, cb, vb = b.correspondences()
        assert np.array_equal(va, vb) and np.array_equal(ca, cb) and np.array_equal(ia, ib), k
        assert a.map_size() == b.map_size() and a.map_size(102) == b.map_size(102) and a.map_size(103) >= 1, k
        guess = Ta
    with pytest.raises(_lib.FlsError):
        k2 = reg.make_matcher("IcpOptimized", reg.YAML_NCLT_ICP)
        k2.ImportMap(blob)  # wrong kind
    a.close(); b.close()


def test_match_batch_equals_fresh_matchers():
    """BASELINE configs[4] (shape): independent scans against ONE map through fls_match_batch on several stream lanes.
    Every job must equal (bit for bit) what a fresh handle computes for the same scan, and agree with a fresh oracle."""
    n_jobs, scale = 7, 0.05
    cfgs = [synth.make_config(1, job=j, scale=scale) for j in range(n_jobs)]
    y = reg.YAML_NCLT_IVOX
    m = reg.make_matcher("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([cfgs[0]["map"]])
    warm = np.eye(4)
    m.Match(reg.PointcloudCluster(planar_cloud_=cfgs[3]["scan"]), warm, update_map=False)  # the owner's own state must not leak into jobs
    clusters = [reg.PointcloudCluster(planar_cloud_=c["scan"]) for c in cfgs]
    oks, Ts, stats = m.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=3)
    oks1, Ts1, stats1 = m.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=1)  # back-to-back path on the owner
    for j, c in enumerate(cfgs):
        f = reg.make_matcher("PointToPlane_IVOX", y)
        f.AddCloudToLocalMap([c["map"]])
        T = np.eye(4)
        ok = f.Match(clusters[j], T, update_map=False)
        assert ok == oks[j] == oks1[j]
        assert np.array_equal(T, Ts[j]) and np.array_equal(T, Ts1[j]), j
        assert f.stats.iterations == stats[j].iterations == stats1[j].iterations
        assert f.stats.n_valid == stats[j].n_valid == stats1[j].n_valid
        o = util.oracle_for("PointToPlane_IVOX", y)
        o.AddCloudToLocalMap(c["map"])
        ok_ref, T_ref = o.Match(c["scan"], np.eye(4), update_map=False)
        dt, dr = synth.pose_error(Ts[j], T_ref)
        assert ok_ref == oks[j] and dt < 1e-4 and dr < 1e-4
        assert o.stats.iterations == stats[j].iterations and o.stats.n_valid == stats[j].n_valid
        f.close()
    m.close()


@pytest.mark.parametrize("mode,y,cid,scale,loc", [("IcpOptimized", reg.YAML_NCLT_ICP, 0, 1.0, True), ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, 0.05, False),
                                                  ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, 3, 0.05, False),
                                                  ("PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, 1, 0.04, True)])
def test_match_batch_other_kinds(mode, y, cid, scale, loc):
    """Stream lanes for every kind: each job equals, bit for bit, what a fresh handle computes (fresh per-job state:
    keyframe gate, valid flags), on 3 lanes and back to back."""
    cfgs = [synth.make_config(cid, job=j, scale=scale) for j in range(4)]
    maps = [cfgs[0]["map"]] + ([cfgs[0]["corner_map"]] if "corner_map" in cfgs[0] else [])
    m = reg.make_matcher(mode, y, is_localization_mode=loc)
    m.AddCloudToLocalMap(maps)
    clusters = [util.cluster_for(mode, c["scan"], c.get("corner_scan")) for c in cfgs]
    oks, Ts, stats = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=3)
    oks1, Ts1, stats1 = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=1)
    for j, c in enumerate(cfgs):
        f = reg.make_matcher(mode, y, is_localization_mode=loc)
        f.AddCloudToLocalMap(maps)
        T = np.eye(4)
        ok = f.Match(clusters[j], T, update_map=False)
        assert ok == oks[j] == oks1[j], (mode, j)
        assert np.array_equal(T, Ts[j]) and np.array_equal(T, Ts1[j]), (mode, j)
        assert f.stats.iterations == stats[j].iterations == stats1[j].iterations and f.stats.n_valid == stats[j].n_valid
        f.close()
    m.close()


def test_match_batch_concurrent_source_filters_stress():
    """Three and four stream lanes of IcpOptimized at once, twenty fresh handles in a row: every lane runs its own device VoxelGrid (exact sort: a persistent
    task kernel per lane, co-resident on the device) before its iterations.  Every pose must equal, bit for bit, what a fresh handle computes alone.
    Round 6: a fence-free hand-over inside that sort (write-through records) passed every single-stream test and the fuzz, and gave ~15 % of these jobs a
    source cloud with an unsorted piece (n_source off by a hundred leaves, pose off by millimetres) -- found by this sequence, not by the one-shot test above."""
    cfgs = [synth.make_config(0, job=j, scale=1.0) for j in range(4)]
    maps = [cfgs[0]["map"]]
    clusters = [util.cluster_for("IcpOptimized", c["scan"], None) for c in cfgs]
    good = []
    for j in range(4):
        f = reg.make_matcher("IcpOptimized", reg.YAML_NCLT_ICP, is_localization_mode=True)
        f.AddCloudToLocalMap(maps)
        T = np.eye(4)
        f.Match(clusters[j], T, update_map=False)
        good.append((T.copy(), f.stats.iterations, f.stats.n_valid, f.stats.n_source))
        f.close()
    for rep in range(20):
        m = reg.make_matcher("IcpOptimized", reg.YAML_NCLT_ICP, is_localization_mode=True)
        m.AddCloudToLocalMap(maps)
        for lanes in (3, 4):
            oks, Ts, st = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=lanes)
            for j in range(4):
                assert (st[j].iterations, st[j].n_valid, st[j].n_source) == good[j][1:], (rep, lanes, j, st[j].n_source, good[j])
                assert np.array_equal(Ts[j], good[j][0]), (rep, lanes, j)
        m.close()


@pytest.mark.parametrize("mode,y,cid,scale,loc", [("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, 1, 0.1, False), ("IcpOptimized", reg.YAML_NCLT_ICP, 0, 1.0, True),
                                                  ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, 0.1, False), ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, 3, 0.1, False)])
def test_exact_tail_solvers(mode, y, cid, scale, loc, monkeypatch):
    """FLS_TAIL_EXACT=1: every 6x6 system goes through the restated Eigen solvers (full-pivot Householder QR / LU inverse) instead
    of the LDL^T fast path -- same parity assertions, and the pose then agrees with the oracle to 1e-13."""
    monkeypatch.setenv("FLS_TAIL_EXACT", "1")
    cfg = synth.make_config(cid, scale=scale)
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    m, o, T, T_ref = run_pair(mode, y, maps, cfg["scan"], corner=cfg.get("corner_scan"), loc=loc, sets_only_tail=(mode == "PointToPlane_IVOX"))
    dt, dr = synth.pose_error(T, T_ref)
    assert dt < 1e-12 and dr < 1e-12, (dt, dr)


def test_rank_deficient_system_takes_the_exact_solver():
    """A scene that constrains three of the six degrees of freedom (one infinite plane): the normal equations are singular, the
    LDL^T fast path declines (no safely positive pivots) and the restated FullPivHouseholderQR -- whose rank-revealing basic solution
    IS the reference's behaviour here -- runs; product == oracle per iteration as everywhere else."""
    rng = np.random.default_rng(5)
    mp = np.zeros((60000, 3), np.float32)
    mp[:, :2] = rng.uniform(-25, 25, size=(60000, 2))
    mp[:, 2] = rng.normal(0, 0.002, size=60000)
    sc = np.zeros((6000, 3), np.float32)
    sc[:, :2] = rng.uniform(-15, 15, size=(6000, 2))
    sc[:, 2] = 0.05 + rng.normal(0, 0.002, size=6000)
    m, o, T, T_ref = run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [mp], sc, sets_only_tail=True)
    assert m.stats.iterations == o.stats.iterations
    assert abs(T[2, 3] - T_ref[2, 3]) < 1e-9 and np.allclose(T[:2, 3], T_ref[:2, 3], atol=1e-9)  # in-plane translation left at the basic solution's zeros


def test_match_batch_ndt_full_size_lanes_share_the_host_pool():
    """Full-size NDT jobs (115,200-point scans: the exact source VoxelGrid takes the host worker pool) on three lanes at once: one lane
    gets the pool, the others filter sequentially -- every job equals a fresh handle's Match bit for bit."""
    cfgs = [synth.make_config(2, job=j) for j in range(3)]
    m = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT)
    m.AddCloudToLocalMap([cfgs[0]["map"]])
    clusters = [util.cluster_for("IncrementalNDT", c["scan"]) for c in cfgs]
    oks, Ts, stats = m.MatchBatch(clusters, [np.eye(4)] * 3, lanes=3)
    for j in range(3):
        f = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT)
        f.AddCloudToLocalMap([cfgs[0]["map"]])
        T = np.eye(4)
        ok = f.Match(clusters[j], T, update_map=False)
        assert ok == oks[j] and np.array_equal(T, Ts[j]), j
        assert f.stats.iterations == stats[j].iterations and f.stats.n_valid == stats[j].n_valid and f.stats.n_source == stats[j].n_source
        f.close()
    m.close()


@pytest.mark.parametrize("job", list(range(1, 9)))
def test_config2_many_scans_reduced(job):
    """Eight more scans (different ground-truth poses and noise draws) at a 4 % slice of configs[1]: the bit-exact
    correspondence claim is a claim about every query, so it is exercised on more than the one benchmark scan."""
    cfg = synth.make_config(1, job=job, scale=0.04)
    run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [cfg["map"]], cfg["scan"], sets_only_tail=True)


@pytest.mark.parametrize("job", [1, 2, 3])
def test_other_kinds_more_scans_reduced(job):
    cfg = synth.make_config(2, job=job, scale=0.05)
    run_pair("IncrementalNDT", reg.YAML_NCLT_NDT, [cfg["map"]], cfg["scan"])
    cfg = synth.make_config(3, job=job, scale=0.05)
    run_pair("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, [cfg["map"], cfg["corner_map"]], cfg["scan"], corner=cfg["corner_scan"])
    cfg = synth.make_config(0, job=job)
    run_pair("IcpOptimized", reg.YAML_NCLT_ICP, [cfg["map"]], cfg["scan"], loc=True)


def test_loam_sparse_maps_second_search_stage():
    """LOAM feature maps thinned 12x: the 5th neighbour usually lies beyond half the gate radius, so the grid search
    has to run its second stage (the 98 shell cells of the 5x5x5 block) -- results still bit-exact."""
    cfg = synth.make_config(3, scale=0.1)
    m, o, T, T_ref = run_pair("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, [cfg["map"][::12].copy(), cfg["corner_map"][::6].copy()], cfg["scan"],
                              corner=cfg["corner_scan"])
    assert m.stats.n_valid > 50


@pytest.mark.parametrize("mode,y,cid,loc", [("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, 1, False), ("IcpOptimized", reg.YAML_NCLT_ICP, 0, True),
                                            ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False), ("LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, 3, False)])
def test_empty_and_tiny_inputs(mode, y, cid, loc):
    """Edge cases at the boundary: an empty source cloud, a source with fewer points than any gate needs, a scan that sees
    nothing of the map (every query without candidates) -- same return value / iteration count / pose as the oracle."""
    cfg = synth.make_config(cid, scale=0.03 if cid else 1.0)
    maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
    empty = np.zeros((0, 3), np.float32)
    far = (cfg["scan"][:200] + np.float32(5000.0)).astype(np.float32)  # 5 km away from every map point
    for scan, corner in ((empty, empty), (cfg["scan"][:7].copy(), cfg.get("corner_scan", empty)[:3].copy()), (far, far[:20].copy())):
        m = reg.make_matcher(mode, y, is_localization_mode=loc)
        o = util.oracle_for(mode, y, loc)
        m.AddCloudToLocalMap(maps)
        o.AddCloudToLocalMap(*maps)
        T = np.eye(4)
        cr = corner if mode == "LoamFull_KdTree" else None
        if mode == "IcpOptimized" and scan.shape[0] <= 10:
            # CHECK_GT(ordered_cloud_.size(), 10u) aborts the reference process (icp_optimized.h:55): an error status here
            with pytest.raises(_lib.FlsError):
                m.Match(util.cluster_for(mode, scan, cr), T, update_map=False)
            m.close()
            continue
        ok = m.Match(util.cluster_for(mode, scan, cr), T, update_map=False)
        ok_ref, T_ref = o.Match(scan, np.eye(4), src1=cr, update_map=False)
        assert ok == ok_ref, (mode, scan.shape)
        assert m.stats.iterations == o.stats.iterations and m.stats.n_valid == o.stats.n_valid, (mode, scan.shape, m.stats.iterations, o.stats.iterations)
        dt, dr = synth.pose_error(T, T_ref)
        assert (dt < 1e-4 and dr < 1e-4) or (not np.all(np.isfinite(T)) and not np.all(np.isfinite(T_ref))), (mode, scan.shape, dt, dr)
        m.close()


def test_handle_lifecycle_many_cycles():
    """Create / use / destroy handles of every kind repeatedly (streams, pinned mailboxes, batch lanes, events): nothing
    leaks into the next handle, results stay bit-identical."""
    cfg = synth.make_config(1, scale=0.02)
    cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
    first = None
    for cycle in range(12):
        m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        m.AddCloudToLocalMap([cfg["map"]])
        oks, Ts, st = m.MatchBatch([cl, cl, cl], [np.eye(4)] * 3, lanes=2 + cycle % 3)
        T = np.eye(4)
        ok = m.Match(cl, T, update_map=bool(cycle % 2))  # every other cycle also grows the map afterwards
        m.set_profiling(True)
        T2 = np.eye(4)
        m.Match(cl, T2, update_map=False)
        m.kernel_time()
        if first is None:
            first = (ok, T.copy(), Ts[0].copy())
        assert ok == first[0] and np.array_equal(T, first[1]) and np.array_equal(Ts[0], first[2]) and np.array_equal(Ts[1], Ts[2])
        m.close()
        for mode, y, cid, loc in (("IcpOptimized", reg.YAML_NCLT_ICP, 0, True), ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, False)):
            if cycle % 4 == 0:
                c2 = synth.make_config(cid, scale=0.02 if cid else 1.0)
                k = reg.make_matcher(mode, y, is_localization_mode=loc)
                k.AddCloudToLocalMap([c2["map"]])
                k.Match(util.cluster_for(mode, c2["scan"], None), np.eye(4), update_map=False)
                k.close()
