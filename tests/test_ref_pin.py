"""The CPU oracle pinned by the reference's OWN code (VERDICT r1 "parity unpinned").

oracle/_ref/libref.so is built by oracle/ref_shim/Makefile from the reference's registration / iVox / LOAM sources,
compiled VERBATIM where they lie under /root/reference against an include-shadow shim (Eigen / PCL / glog stand-ins whose
third-party arithmetic forwards to the oracle's own flo_linalg / flo_kdtree / voxel_grid).  So these tests pin every
line of the reference's own control flow and data handling -- Gauss-Newton loops, stop rules, stale flags (Q1), stale
neighbour lists (Q15), the iVox insert rule + LRU, std::nth_element slot order, deques, keyframe gates, Q10, Q11,
NDT voxel statistics, the feature walk -- and leave Eigen's association order as the one unpinned thing
(FP64 fields are compared to 1e-9, observed <= 2e-14; everything integer or float-only is compared exactly).

  * live tests: need libref.so (built here whenever /root/reference exists; the .so also travels with a gpurun snapshot)
  * golden tests: tests/golden/ref_*.npz (made by tests/golden/make_ref_golden.py from libref.so) -- run anywhere.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O, ref as R
from tests import refpin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
need_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so absent and /root/reference not present to build it")


@need_ref
@pytest.mark.parametrize("name", refpin.SCENARIOS)
def test_oracle_equals_compiled_reference(name):
    """Multi-scan replay (mapping mode: ivox, ivox_lru, icp, ndt, loam) / localization mode (+ GetFitnessScore) through the compiled
    reference in a fresh process vs the oracle, every frame, every field (tests/refpin.py lists them)."""
    ref_out = refpin.run_ref_subprocess(name)
    ora_out = refpin.run("oracle", name)
    worst = refpin.compare(ora_out, ref_out, name)
    assert worst["T"] < 1e-12, worst


@need_ref
@pytest.mark.parametrize("name", refpin.SCENARIOS)
def test_compiled_reference_parallel_pstl_equals_serial(name):
    """oracle/_ref/libref_par.so (the reference's `std::execution::par / par_unseq` loops on OpenMP threads -- ref_shim/include/pstl_omp.hpp
    in place of the TBB backend the reference links; what bench.py times as the multi-core cpu_baseline_ref) against the serial build: every
    frame, every field, every bit -- the parallel loops write per-index outputs only."""
    ser = refpin.run_ref_subprocess(name)
    par = refpin.run_ref_subprocess(name, env={"FLS_REF_PAR": "1", "OMP_NUM_THREADS": "4"})
    worst = refpin.compare(par, ser, name)
    assert all(v == 0.0 for v in worst.values()), worst


@need_ref
@pytest.mark.parametrize("seed", list(range(24)))
def test_oracle_equals_compiled_reference_fuzz(seed):
    """Seeded random variations of the mapping-mode replays (tests/refpin.py::make_fuzz_scenario: kind by seed % 4; iteration budgets of two,
    gates far tighter / looser than the YAML's, deques of one or two frames, an effective-point floor the scan cannot meet, iVox LRU capacities
    of a few hundred voxels, start poses anywhere in the room): the oracle follows the compiled reference through every frame and field.
    (120 seeds were run once: profiles/r05_ref_pin_fuzz_200_scenarios.log; what the draw had to avoid is a reference crash, see the generator.)"""
    name = f"fuzz{seed}"
    ref_out = refpin.run_ref_subprocess(name)
    ora_out = refpin.run("oracle", name)
    worst = refpin.compare(ora_out, ref_out, name)
    assert worst["T"] < 1e-12, worst


DEGENERATE = ("deg_ivox_empty", "deg_ivox_tiny", "deg_ivox_far", "deg_icp_tiny12", "deg_icp_far", "deg_ndt_empty", "deg_ndt_tiny", "deg_ndt_far",
              "deg_loam_nocorner", "deg_loam_tiny", "deg_loam_far", "deg_kd_tiny", "deg_kd_far")


@need_ref
@pytest.mark.parametrize("name", DEGENERATE)
def test_oracle_equals_compiled_reference_degenerate_inputs(name):
    """Empty source clouds, sources with fewer points than any gate needs, scans that see nothing of the map (every query without candidates),
    LoamFull without corner features: return value, pose, counts and map state of the oracle are the compiled reference's."""
    worst = refpin.compare(refpin.run("oracle", name), refpin.run_ref_subprocess(name), name)
    assert worst["T"] < 1e-10, worst  # (thirty iterations on a rank-deficient 6 x 6 system, deg_loam_tiny, amplify the last bits to 4e-12)


@need_ref
def test_compiled_reference_aborts_on_ten_or_fewer_icp_points():
    """CHECK_GT(ordered_cloud_.size(), 10u) (icp_optimized.h:55) ends the reference process; the product returns an error status for it
    (tests/test_gpu_parity.py::test_empty_and_tiny_inputs)."""
    import subprocess
    with pytest.raises(subprocess.CalledProcessError):
        refpin.run_ref_subprocess("deg_icp_le10")


@need_ref
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_equals_compiled_reference_fuzz_long_runs(seed):
    """8-14 frames per scenario (one seed per kind here, 60 run once): deques past the length at which the reference VoxelGrids them, LRU lists at
    capacity for many frames, keyframe gates passing and failing in turn."""
    name = f"fuzzL{seed}"
    worst = refpin.compare(refpin.run("oracle", name), refpin.run_ref_subprocess(name), name)
    assert worst["T"] < 1e-12, worst


@need_ref
@pytest.mark.parametrize("seed", list(range(12)))
def test_oracle_equals_compiled_reference_fuzz_localization(seed):
    """The same for localization mode (kind by seed % 4: IcpOptimized, LoamPointToPlaneKdtree, LoamPointToPlaneIVOX, IncrementalNDT): prior maps of
    random size, random thresholds / budgets, GetFitnessScore after every Match (40 seeds run once, all equal: the same log file)."""
    name = f"lfuzz{seed}"
    ref_out = refpin.run_ref_subprocess(name)
    ora_out = refpin.run("oracle", name)
    worst = refpin.compare(ora_out, ref_out, name)
    assert worst["T"] < 1e-12, worst


@need_ref
@pytest.mark.parametrize("seed", list(range(12)))
def test_feature_oracle_equals_compiled_reference_fuzz(seed):
    """PointcloudProjector + FeatureExtractor on seeded random frames (either lidar model, tilted poses, thresholds and range gates away from the
    YAML's, up to 90 % of the returns dropped, driver order shuffled or not): every array bit for bit (40 seeds run once, all equal)."""
    name = f"ffuzz{seed}"
    ref_out = refpin.run_features_subprocess(name)
    ora = refpin.run_features("oracle", name, sort_mode=1)
    assert ref_out["digest"] == ora["digest"]
    assert ref_out["n_ordered"] == ora["n_ordered"] and ref_out["extracted"] == ora["extracted"]
    for f in refpin.FEATURE_FIELDS:
        assert ref_out[f].shape == ora[f].shape and np.array_equal(ref_out[f], ora[f]), (name, f)


@pytest.mark.parametrize("name", refpin.SCENARIOS)
def test_oracle_equals_reference_golden(name):
    g = refpin.from_golden(np.load(os.path.join(GOLD, f"ref_{name}.npz")))
    ora_out = refpin.run("oracle", name)
    if ora_out["digest"] != g["digest"]:
        pytest.skip("numpy on this host generates a different synthetic scenario than the one the golden was made from")
    refpin.compare(ora_out, g, name)


@need_ref
@pytest.mark.parametrize("name", list(refpin.FEATURE_CASES))
def test_feature_oracle_equals_compiled_reference(name):
    """PointcloudProjector::Project + FeatureExtractor::ExtractFeatures, every array bit for bit -- including the order of
    equal-roughness points, which is libstdc++'s std::sort permutation in the reference (oracle sort_mode 1)."""
    ref_out = refpin.run_features_subprocess(name)
    ora = refpin.run_features("oracle", name, sort_mode=1)
    assert ref_out["n_ordered"] == ora["n_ordered"] and ref_out["extracted"] == ora["extracted"]
    for f in refpin.FEATURE_FIELDS:
        assert ref_out[f].shape == ora[f].shape and np.array_equal(ref_out[f], ora[f]), (name, f)


@pytest.mark.parametrize("name", list(refpin.FEATURE_CASES))
def test_feature_oracle_equals_reference_golden_and_tie_order_is_the_only_freedom(name):
    g = np.load(os.path.join(GOLD, f"ref_features_{name}.npz"))
    ora = refpin.run_features("oracle", name, sort_mode=1)
    if ora["digest"] != str(g["digest"]):
        pytest.skip("different synthetic scan on this host")
    for f in refpin.FEATURE_FIELDS:
        assert refpin._sha(ora[f]) == str(g[f]), (name, f)
    # the oracle's default (ties in index order = what the HIP kernel reproduces) differs from the reference's libstdc++ order
    # only by a permutation inside groups of EXACTLY equal roughness: same selections, same planar set
    st = refpin.run_features("oracle", name, sort_mode=0)
    for f in ("ordered", "depth", "col", "row_start", "row_end", "corner", "is_corner", "valid_post"):
        assert np.array_equal(st[f], ora[f]), (name, f)
    assert np.array_equal(refpin._sorted_rows(st["planar"]), refpin._sorted_rows(ora["planar"]))
    moved = int((st["planar"] != ora["planar"]).any(1).sum())
    raw, params = refpin.make_feature_case(name)
    o = O.OracleFeatures(**params)
    o.Project(raw); o.ExtractFeatures()
    assert moved <= 2 * o.tie_pairs() + 2, (moved, o.tie_pairs())


@need_ref
def test_small_functions_against_compiled_reference():
    """SO3Exp / RotationMatrixToRPY (math_function.h), FastAtan2, LidarModel::ColIndex: oracle vs the reference's compiled code."""
    rng = np.random.default_rng(5)
    L = R.lib()
    for _ in range(200):
        v = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 0.3, 2.5])
        a = np.zeros(9); b = np.zeros(9)
        L.ref_so3_exp(v.ctypes.data_as(O.C.POINTER(O.C.c_double)), a.ctypes.data_as(O.C.POINTER(O.C.c_double)))
        Rm = O.so3_exp(v)
        assert np.allclose(a.reshape(3, 3, order="F"), Rm, rtol=0, atol=1e-15)
        rr = np.zeros(3)
        L.ref_rpy(a.ctypes.data_as(O.C.POINTER(O.C.c_double)), rr.ctypes.data_as(O.C.POINTER(O.C.c_double)))
        assert np.allclose(rr, O.rpy(a.reshape(3, 3, order="F")), rtol=0, atol=1e-15)
    xy = rng.normal(size=(2000, 2)).astype(np.float32) * 30
    for x, y in xy:
        assert L.ref_fast_atan2f(float(y), float(x)) == O.lib().flo_fast_atan2f(float(y), float(x))


@need_ref
def test_ivox_eviction_and_recreation_inside_one_batch_against_compiled_reference():
    """IVoxMap::AddPoints evicts inside its insert loop (ivox_map.cpp:133-136): one 3,000-point batch at capacity 60 with 722 evictions, 641 of them of
    voxels a later point of the same batch re-creates -- the corner the device-side eviction walk resolves (kernels_ivox_update.hpp).  The reference's own
    code (fresh process), a sequential Python model and the oracle agree on the LRU order, every voxel's points and the surviving points (tests/evict_pin.py)."""
    from tests import evict_pin as E
    n_evict, n_recreate = E.check(E.run_ref_subprocess(), E.run_oracle())
    assert (n_evict, n_recreate) == (722, 641)


def test_ivox_eviction_and_recreation_inside_one_batch_against_reference_golden():
    """the same check against the committed output of the compiled reference (tests/golden/ref_ivox_recreate.npz, written by `python -m tests.evict_pin --golden`)"""
    from tests import evict_pin as E
    with np.load(E.GOLDEN) as z:
        ref = {k: z[k] for k in z.files}
    E.check(ref, E.run_oracle())
