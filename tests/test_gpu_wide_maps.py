"""Maps that do not fit a dense bounding-box window, and the table-lookup paths next to the dense ones.

The reference's IVoxMap bounds the number of alive voxels and nothing else (include/ivox_map/ivox_map.h:35,
src/ivox_map/ivox_map.cpp:122-147): a mapping run may wander for kilometres.  These tests drive the HIP path through maps whose
occupied voxels span far more than any dense window (several sites up to 2.5 km apart), and through the A/B switches that select
the table form of a lookup, with the same assertions as the dense tests: return values, iteration counts, n_valid per iteration,
flags, neighbour counts, correspondence ids, poses, map sizes -- all against the CPU oracle."""
import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth
from tests import replay, util
from tests.test_gpu_parity import run_pair, _replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


def site_offset(s: int) -> np.ndarray:
    """site s of a multi-site world: 500 m apart along a diagonal (float32 coordinates stay exact to ~1e-4 m at 2.5 km)"""
    return np.array([400.0 * s, 300.0 * s, 4.0 * s])


def shifted(cloud: np.ndarray, off: np.ndarray) -> np.ndarray:
    return (cloud.astype(np.float64) + off[None, :]).astype(np.float32)


@pytest.mark.parametrize("scale", [0.05, 1.0])
def test_config2_p2plane_ivox_table_lookup(scale, monkeypatch):
    """BASELINE configs[1] with FLS_IVOX_DENSE=0 (the per-voxel hash table instead of the two-level brick image)."""
    monkeypatch.setenv("FLS_IVOX_DENSE", "0")
    cfg = synth.make_config(1, scale=scale)
    m, o, T, T_ref = run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [cfg["map"]], cfg["scan"], sets_only_tail=True)
    assert m.stats.converged == 1


@pytest.mark.parametrize("variant,balanced", [("8", "1"), ("4", "0"), ("8", "0")])
def test_config2_p2plane_ivox_kernel_variants(variant, balanced, monkeypatch):
    """The other instantiations of the correspondence kernel (8 lanes per query, whole voxels per lane) give the same lists."""
    monkeypatch.setenv("FLS_IVOX_VARIANT", variant)
    monkeypatch.setenv("FLS_IVOX_BALANCED", balanced)
    cfg = synth.make_config(1, scale=0.1)
    run_pair("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, [cfg["map"]], cfg["scan"], sets_only_tail=True)


def test_mapping_replay_table_lookup(monkeypatch):
    """8-scan mapping replay with the hash-table image: the map update runs on the exact host path, every Match equals the oracle."""
    monkeypatch.setenv("FLS_IVOX_DENSE", "0")
    m, o = _replay(8)
    assert m.map_size(103) == 0, "the table image is maintained by the host path"
    assert m.map_size() > 60000


def test_match_batch_table_lookup(monkeypatch):
    monkeypatch.setenv("FLS_IVOX_DENSE", "0")
    n_jobs = 5
    cfgs = [synth.make_config(1, job=j, scale=0.05) for j in range(n_jobs)]
    y = reg.YAML_NCLT_IVOX
    m = reg.make_matcher("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([cfgs[0]["map"]])
    clusters = [reg.PointcloudCluster(planar_cloud_=c["scan"]) for c in cfgs]
    oks, Ts, stats = m.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=3)
    for j, c in enumerate(cfgs):
        o = util.oracle_for("PointToPlane_IVOX", y)
        o.AddCloudToLocalMap(cfgs[0]["map"])
        ok_ref, T_ref = o.Match(c["scan"], np.eye(4), update_map=False)
        dt, dr = synth.pose_error(Ts[j], T_ref)
        assert ok_ref == oks[j] and dt < 1e-12 and dr < 1e-12, (j, dt, dr)
        assert o.stats.iterations == stats[j].iterations and o.stats.n_valid == stats[j].n_valid
        o.close()
    m.close()


def multi_site_replay(n_sites: int, n_frames: int, capacity=None, monkeypatch=None):
    """A mapping run that visits `n_sites` sites 500 m apart in turn: the prior map holds a patch around every site, frame k is
    taken at site k % n_sites (its own pose chain) and grows the map there.  The occupied voxels span n_sites x (800 x 600) cells."""
    scene = synth.make_scene()
    rng = synth.rng_for(1, 41)
    radius = 26.0
    patch = synth.sample_map(scene, 40000, synth.rng_for(1, 0, 7), radius=radius)
    prior = np.concatenate([shifted(patch, site_offset(s)) for s in range(n_sites)])
    lid = dict(synth.VELODYNE_64, n_az=50)
    y = reg.YAML_NCLT_IVOX
    o = util.oracle_for("PointToPlane_IVOX", y)
    if capacity is not None:
        monkeypatch.setenv("FLS_IVOX_CAPACITY", str(capacity))
        o.set_ivox_capacity(capacity)
    m = reg.make_matcher("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([prior])
    o.AddCloudToLocalMap(prior)
    assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels()
    local_gt = [np.eye(4) for _ in range(n_sites)]
    guess = []
    for s in range(n_sites):
        W = np.eye(4)
        W[:3, 3] = site_offset(s)
        guess.append(W)
    for k in range(n_frames):
        s = k % n_sites
        local_gt[s] = local_gt[s] @ synth.random_pose(rng, 1.0, 0.6)
        local_gt[s][2, 3] = rng.uniform(-0.06, 0.06)  # (the height must not random-walk into the ground over many frames)
        scan = synth.cast_scan(scene, local_gt[s], rng=rng, max_range=radius + 8.0, **lid)  # sees beyond the patch: the map grows
        T = guess[s].copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=scan), T, update_map=True)
        ok_ref, T_ref = o.Match(scan, guess[s], update_map=True)
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        assert ok_ref, (k, s)
        assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels(), (k, m.map_size(), o.map_size(), m.map_size(102), o.map_voxels())
        guess[s] = T_ref
    return m, o


def test_multi_site_mapping_run_2km_extent():
    """Six sites, 2.5 km end to end (bounding box 4,000 x 3,000 x 90 voxels = 1.1e9 cells): the map stays on the device -- the brick
    image has no extent limit -- and every one of 24 scans equals the oracle."""
    m, o = multi_site_replay(6, 24)
    applied, refused = m.map_size(103), m.map_size(104)
    print(f"multi-site run: {applied} device batches, {refused} refused (conflict {m.map_size(119)}, full {m.map_size(120)}, outside {m.map_size(121)}), "
          f"{m.map_size(102)} voxels, image rebuilds {m.map_size(101)}")
    assert applied >= 23 and refused == 0, (applied, refused)
    assert m.map_size(101) == 1, "only the initial build flattens the map"
    m.close(); o.close()


def test_multi_site_mapping_run_with_evictions(monkeypatch):
    """The same run with the LRU capacity just above the prior map: sites visited least recently lose voxels while the current one grows;
    evictions run inside the device batches, including voxels evicted and re-created by one batch (eviction-order conflicts; round 3 replayed those on the host)."""
    probe = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    scene = synth.make_scene()
    patch = synth.sample_map(scene, 40000, synth.rng_for(1, 0, 7), radius=26.0)
    probe.AddCloudToLocalMap(np.concatenate([shifted(patch, site_offset(s)) for s in range(4)]))
    cap = probe.map_voxels() + 300
    probe.close()
    m, o = multi_site_replay(4, 20, capacity=cap, monkeypatch=monkeypatch)
    applied, refused, evicted = m.map_size(103), m.map_size(104), m.map_size(117)
    print(f"multi-site run at capacity {cap}: {applied} device batches, {refused} refused (conflict {m.map_size(119)}, full {m.map_size(120)}, outside {m.map_size(121)}), {evicted} evicted on the device")
    assert o.map_voxels() == cap - 1
    print(f"voxels evicted and re-created inside one device batch: {m.map_size(126)}")
    assert refused == 0 and m.map_size(119) == 0 and m.map_size(120) == 0 and m.map_size(121) == 0, "nothing leaves the device: eviction-order conflicts are resolved there"
    assert applied >= 10 and evicted > 200, (applied, evicted)
    m.close(); o.close()


@pytest.mark.parametrize("mode,y,cid", [("IcpOptimized", reg.YAML_NCLT_ICP, 0), ("PointToPlane_KdTree", reg.YAML_NCLT_LOC_KDTREE, 1)])
def test_kd_kinds_on_a_map_wider_than_any_window(mode, y, cid):
    """Localization-mode kd kinds against a map made of two copies of the scene 2.5 km apart (the cell grid's bounding box exceeds the
    dense-window budget: the search runs on the hash-table form of the grid).  The scan is registered at the FAR site."""
    cfg = synth.make_config(cid, scale=1.0 if cid == 0 else 0.04)
    off = site_offset(5)
    wide = np.concatenate([cfg["map"], shifted(cfg["map"], off)])
    W = np.eye(4)
    W[:3, 3] = off
    m, o, T, T_ref = run_pair(mode, y, [wide], cfg["scan"], loc=True, T_init=W)
    assert m.GetFitnessScore(2.0) == pytest.approx(o.GetFitnessScore(2.0), rel=1e-6)
    dt, dr = synth.pose_error(T, W @ cfg["T_gt"])
    assert dt < 0.25 and dr < 0.01, (dt, dr)  # (a 4 % slice of the scan: it registers, to the accuracy such a slice gives)


def test_ndt_on_a_multi_site_map():
    """IncrementalNDT keys its voxels through a hash table at any extent: two sites 2.5 km apart, mapping mode, device updates."""
    cfg = synth.make_config(2, scale=0.1)
    off = site_offset(5)
    wide = np.concatenate([cfg["map"], shifted(cfg["map"], off)])
    W = np.eye(4)
    W[:3, 3] = off
    m = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT)
    o = util.oracle_for("IncrementalNDT", reg.YAML_NCLT_NDT)
    m.AddCloudToLocalMap([wide])
    o.AddCloudToLocalMap(wide)
    for k, guess in enumerate((np.eye(4), W, np.eye(4), W)):
        T = guess.copy()
        ok = m.Match(util.cluster_for("IncrementalNDT", cfg["scan"]), T, update_map=True)
        ok_ref, T_ref = o.Match(cfg["scan"], guess, update_map=True)
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref)
        assert m.map_size() == o.map_size(), k
    m.close(); o.close()


@pytest.mark.parametrize("form", ["short", "long", "one_workgroup"])
def test_device_addpoints_three_forms(form, monkeypatch):
    """The device AddPoints runs the same device functions in three launch structures: the SHORT chain (default: the decision launch
    counts, every block derives its offsets and the verdict itself, the last block to finish publishes -- five launches behind the
    decision instead of ten), the round-3 LONG chain (FLS_IVOX_SHORT_CHAIN=0; what batches that may evict still use) and ONE launch of
    one workgroup (FLS_IVOX_FUSED_UPDATE=1).  The 8-scan mapping replay equals the oracle in all of them."""
    if form == "long":
        monkeypatch.setenv("FLS_IVOX_SHORT_CHAIN", "0")
    if form == "one_workgroup":
        monkeypatch.setenv("FLS_IVOX_FUSED_UPDATE", "1")
    m, o = _replay(8)
    assert m.map_size(103) >= 7 and m.map_size(104) == 0
    short, one = m.map_size(123), m.map_size(122)
    # (short-chain launches include the speculative ones the device skipped: map_size(125))
    assert (short - m.map_size(125), one) == ((m.map_size(103), 0) if form == "short" else (0, m.map_size(103)) if form == "one_workgroup" else (0, 0)), (form, short, one)
    m.close(); o.close()


def test_device_addpoints_large_and_small_batches_alternate(monkeypatch):
    """Raw 64 x 600 scans (38,400 points, beyond the one-workgroup limit: short chain) alternate with small ones (one workgroup, switched on here)."""
    monkeypatch.setenv("FLS_IVOX_FUSED_UPDATE", "1")
    scene = synth.make_scene()
    rng = synth.rng_for(1, 77)
    mp = synth.sample_map(scene, 120000, synth.rng_for(1, 0, 8), radius=40.0)
    y = reg.YAML_NCLT_IVOX
    m = reg.make_matcher("PointToPlane_IVOX", y)
    o = util.oracle_for("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([mp])
    o.AddCloudToLocalMap(mp)
    Tgt, guess = np.eye(4), np.eye(4)
    for k, n_az in enumerate((600, 60, 600, 60)):
        Tgt = Tgt @ synth.random_pose(rng, 1.0, 0.5)
        Tgt[2, 3] = 0.0
        scan = synth.cast_scan(scene, Tgt, rng=rng, max_range=48.0, **dict(synth.VELODYNE_64, n_az=n_az))
        T = guess.copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=scan), T, update_map=True)
        ok_ref, T_ref = o.Match(scan, guess, update_map=True)
        util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
        assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels(), k
        guess = T_ref
    assert m.map_size(103) == 4 and m.map_size(122) == 2 and m.map_size(123) == 2 and m.map_size(104) == 0, (m.map_size(103), m.map_size(122), m.map_size(123), m.map_size(104))
    m.close(); o.close()


@pytest.mark.parametrize("spec", ["1", "0"])
def test_speculative_update_chain(spec, monkeypatch):
    """The decision + update chain is queued behind the iterations the Match is expected to need and gates itself on the device
    (Gauss-Newton loop ended and n_valid >= 50: what LoamPointToPlaneIVOX::Match checks before AddCloudToLocalMap, :197-206); a chain that
    finds the Match unfinished skips itself and the host queues another one behind the added iterations.  FLS_IVOX_SPECULATIVE=0 waits for
    the result first.  Same poses, ids, map sizes as the oracle either way (the 8-scan replay + the 24-scan multi-site run)."""
    monkeypatch.setenv("FLS_IVOX_SPECULATIVE", spec)
    m, o = _replay(8)
    assert m.map_size(103) >= 7 and m.map_size(104) == 0
    queued, skipped = m.map_size(124), m.map_size(125)
    print(f"speculative = {spec}: {queued} chains queued, {skipped} skipped on the device, {m.map_size(103)} batches applied")
    assert (queued >= 7 and queued - skipped == m.map_size(103)) if spec == "1" else (queued == 0 and skipped == 0), (queued, skipped, m.map_size(103))
    m.close(); o.close()
    m, o = multi_site_replay(3, 9)
    assert m.map_size(104) == 0
    m.close(); o.close()


def test_speculative_chain_skips_itself_when_the_match_does_not_converge():
    """A scan that sees nothing of the map: n_valid = 0 < 50, Match returns false, the reference does not touch the map -- the chain queued
    behind the iterations must leave the image alone (status kUpdSkipped), and the next ordinary scan must still agree with the oracle."""
    scene = synth.make_scene()
    rng = synth.rng_for(1, 55)
    mp = synth.sample_map(scene, 60000, synth.rng_for(1, 0, 5), radius=30.0)
    y = reg.YAML_NCLT_IVOX
    m = reg.make_matcher("PointToPlane_IVOX", y)
    o = util.oracle_for("PointToPlane_IVOX", y)
    m.AddCloudToLocalMap([mp]); o.AddCloudToLocalMap(mp)
    lid = dict(synth.VELODYNE_64, n_az=60)
    good = synth.cast_scan(scene, np.eye(4), rng=rng, max_range=38.0, **lid)
    far = (good[:3000] + np.float32(4000.0)).astype(np.float32)
    guess = np.eye(4)
    for k, scan in enumerate((good, far, good)):
        T = guess.copy()
        ok = m.Match(reg.PointcloudCluster(planar_cloud_=scan), T, update_map=True)
        ok_ref, T_ref = o.Match(scan, guess, update_map=True)
        assert ok == ok_ref and m.stats.map_updated == o.stats.map_updated, (k, ok, ok_ref)
        assert m.map_size() == o.map_size() and m.map_size(102) == o.map_voxels(), k
        if ok_ref:
            util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, sets_only_tail=True, max_tie_rows=int(o.counters().tie_queries))
    assert m.map_size(125) >= 1, "the far scan's chain must have skipped itself"
    m.close(); o.close()
