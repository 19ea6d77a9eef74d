"""Long mapping-mode trajectories, the Gauss-Newton tail's SPD fast path (LDL^T in one lane, the default) against the restated
Eigen solvers (FLS_TAIL_EXACT=1: FullPivHouseholderQR for the LOAM family, partial-pivot LU inverse for ICP / NDT) -- VERDICT r2
weak #2 / next #1d, ADVICE r2 (kernels_p2plane.hpp:188).

The fast path solves the same 6x6 system, so a step differs from the reference arithmetic's by ~1e-13 relative.  What has to
be shown is that over a LONG run (>= 100 scans, map growing through Match itself: insert rule + LRU, deques + keyframe gates,
NDT voxel statistics) that perturbation never flips a discrete decision: per scan the two handles must agree on the return
value, the iteration count, n_valid (planar / corner), the map_updated (keyframe) decision, every map size, every valid flag,
neighbour count and correspondence id -- and the poses must stay within 1e-9 (they are compared after EVERY scan; each handle
follows its own pose chain, so an accumulated drift would show).
"""
import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth
from tests import replay, util

pytestmark = pytest.mark.gpu

N_FRAMES = 100


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


@pytest.mark.parametrize("name", ["ivox", "icp", "ndt_dev", "loam"])
def test_ldlt_fast_path_never_flips_a_decision_over_100_scans(name, monkeypatch):
    r = replay.make_replay(name, n_frames=N_FRAMES, yaw_long_deg=9.0)
    mode, y = r["mode"], r["y"]
    if name == "icp":
        y = dict(y, optimization_iter_num=30)  # the short scenario's 12 is there to provoke Q10; here the YAML value
    monkeypatch.setenv("FLS_TAIL_EXACT", "0")
    fast = reg.make_matcher(mode, y)
    monkeypatch.setenv("FLS_TAIL_EXACT", "1")  # read once, when the handle is created (matcher_base.hpp::init_common)
    exact = reg.make_matcher(mode, y)
    monkeypatch.delenv("FLS_TAIL_EXACT")
    slots = (0, 1) if mode == "LoamFull_KdTree" else (0,)
    for m in (fast, exact):
        m.AddCloudToLocalMap(r["init_clouds"])
    Tp = [np.eye(4), np.eye(4)]
    worst_dt = worst_dr = 0.0
    n_upd = n_ok = 0
    for k, f in enumerate(r["frames"]):
        res = []
        for j, m in enumerate((fast, exact)):
            T = Tp[j] @ f["guess_step"]
            ok = m.Match(util.cluster_for(mode, f["scan"], f["corner"]), T, update_map=True)
            Tp[j] = T
            res.append((ok, int(m.stats.iterations), int(m.stats.n_valid), int(m.stats.n_valid_corner), int(m.stats.map_updated), int(m.stats.n_source),
                        tuple(m.map_size(s) for s in slots)))
        assert res[0] == res[1], (name, k, res)
        for s in slots:
            a, b = fast.correspondences(s), exact.correspondences(s)
            for u, v, what in zip(a, b, ("ids", "cnt", "valid")):
                assert np.array_equal(u, v), (name, k, s, what, int((u != v).sum()))
        dt, dr = synth.pose_error(Tp[0], Tp[1])
        worst_dt, worst_dr = max(worst_dt, dt), max(worst_dr, dr)
        assert dt <= 1e-9 and dr <= 1e-9, (name, k, dt, dr)
        n_upd += res[0][4]
        n_ok += int(res[0][0])
    print(f"{name}: {N_FRAMES} scans, {n_ok} converged, {n_upd} map updates, worst |dt| {worst_dt:.2e} m |dR| {worst_dr:.2e} rad between the two tails")
    assert n_upd >= N_FRAMES // 4, "the trajectory must keep growing the map"
    for m in (fast, exact):
        m.close()
