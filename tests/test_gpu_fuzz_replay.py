"""Seeded random mapping-mode scenarios on the GPU (tests/refpin.py::make_fuzz_scenario: parameters far from the YAML sets, start poses anywhere in
the room, 3-5 frames, iVox LRU capacities of a few hundred voxels) -- the scenarios on which the oracle equals the reference's own compiled code
(tests/test_ref_pin.py::test_oracle_equals_compiled_reference_fuzz; profiles/r05_ref_pin_fuzz_*_scenarios.log).  After every frame: return value,
iteration count, per-iteration n_valid / pose, flags, counts, ids, map_updated, map sizes (tests/test_gpu_mapping_replay.py::run_replay).
One seed per kind here (the four that ran on a GPU in round 5); tools/gpu_fuzz_replay.py runs any range of seeds."""
import pytest

from funny_lidar_slam_amd import _lib
from tests import refpin, util
from tests.test_gpu_mapping_replay import run_replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fuzz_scenario_equals_oracle(seed, monkeypatch, built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"
    sc = refpin.make_fuzz_scenario(seed)
    if "ivox_capacity" in sc:  # the LRU capacity is a constructor constant of the reference (ivox_map.h); the handle takes it through its test hook
        cap = sc["ivox_capacity"]
        monkeypatch.setenv("FLS_IVOX_CAPACITY", str(cap))
        orig = util.oracle_for

        def with_cap(mode, y, loc=False):
            o = orig(mode, y, loc)
            o.set_ivox_capacity(cap)
            return o
        monkeypatch.setattr(util, "oracle_for", with_cap)
    r, hist = run_replay(sc["name"], r=sc)
    assert len(hist) == len(sc["frames"])
