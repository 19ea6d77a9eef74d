"""HIP path vs the CPU oracle on ANY scenario of tests/refpin.py::make_scenario -- the scenarios on which the oracle was pinned against the
reference's own compiled code (tests/test_ref_pin.py): the seeded mapping-mode replays (`fuzz<seed>`), the seeded localization-mode runs with
GetFitnessScore after every Match (`lfuzz<seed>`) and the degenerate inputs (`deg_*`).  Test infrastructure (VERDICT r5 missing #4): the
assertions per frame are those of tests/test_gpu_mapping_replay.py::run_replay -- return value, iteration count, per-iteration n_valid / pose /
residual sums, valid flags, neighbour counts and ids, map_updated, map sizes -- plus the fitness score in localization mode.
Reference behaviours the scenarios reach: icp_optimized.h:55 (CHECK_GT(size, 10) aborts -> an error status here), :151-162 (Q10),
incremental_ndt.h:306-309 (effective-point floor), loam_point_to_plane_ivox.h:226-228 (FloatNaN outside localization mode)."""
from __future__ import annotations

import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth
from tests import refpin, util


def run_scenario(name: str, sc: dict = None) -> list:
    """Drive the HIP matcher and the oracle through one scenario; raises AssertionError on the first difference.  Returns the per-frame history."""
    sc = sc if sc is not None else refpin.make_scenario(name)
    mode, y, loc = sc["mode"], sc["y"], bool(sc.get("loc", False))
    cap = sc.get("ivox_capacity")
    import os
    prev_cap = os.environ.get("FLS_IVOX_CAPACITY")
    if cap is not None:  # the LRU capacity is a constructor constant of the reference (ivox_map.h); the handle takes it through its test hook
        os.environ["FLS_IVOX_CAPACITY"] = str(cap)
    try:
        m = reg.make_matcher(mode, y, is_localization_mode=loc) if mode != "LoamFull_KdTree" else reg.make_matcher(mode, y)
    finally:
        if cap is not None:
            if prev_cap is None:
                os.environ.pop("FLS_IVOX_CAPACITY", None)
            else:
                os.environ["FLS_IVOX_CAPACITY"] = prev_cap
    o = util.oracle_for(mode, y, loc)
    if cap is not None:
        o.set_ivox_capacity(cap)
    try:
        m.AddCloudToLocalMap(sc["init_clouds"])
        o.AddCloudToLocalMap(*sc["init_clouds"])
        slots = (0, 1) if mode == "LoamFull_KdTree" else (0,)
        for s in slots:
            assert m.map_size(s) == o.map_size(s), (name, "init", s, m.map_size(s), o.map_size(s))
        Tprev = np.eye(4)
        hist = []
        for k, f in enumerate(sc["frames"]):
            guess = f["absolute_guess"] if "absolute_guess" in f else Tprev @ f["guess_step"]
            T = np.array(guess, dtype=np.float64).copy()
            cl = util.cluster_for(mode, f["scan"], f["corner"])
            if mode == "IcpOptimized" and f["scan"].shape[0] <= 10:
                # CHECK_GT(ordered_cloud_.size(), 10u) aborts the reference process (icp_optimized.h:55; tests/test_ref_pin.py pins the abort): an error status here
                with pytest.raises(_lib.FlsError):
                    m.Match(cl, T, update_map=True)
                hist.append(dict(ok=None, upd=0, iters=0, abort=True))
                continue
            ok = m.Match(cl, T, update_map=True)
            ok_ref, T_ref = o.Match(f["scan"], np.array(guess, dtype=np.float64), src1=f["corner"], update_map=True)
            finite = bool(np.all(np.isfinite(T_ref)))
            if finite:
                ties = int(o.counters().tie_queries)
                util.assert_same_registration(m, o, ok, T, ok_ref, T_ref, slots=slots, sets_only_tail=(mode == "PointToPlane_IVOX"), max_tie_rows=ties)
            else:  # a solve on an all-zero system (no valid point): NaN poses on both sides, everything countable still has to agree
                assert ok == ok_ref and not np.all(np.isfinite(T)), (name, k)
                assert m.stats.iterations == o.stats.iterations and m.stats.n_valid == o.stats.n_valid and m.stats.n_valid_corner == o.stats.n_valid_corner, (name, k)
            assert m.stats.map_updated == o.stats.map_updated, (name, k, m.stats.map_updated, o.stats.map_updated)
            assert m.stats.n_source == o.stats.n_source and m.stats.n_source_corner == o.stats.n_source_corner, (name, k)
            for s in slots:
                assert m.map_size(s) == o.map_size(s), (name, k, s, m.map_size(s), o.map_size(s))
            rec = dict(ok=bool(ok_ref), upd=int(o.stats.map_updated), size=o.map_size(0), iters=int(o.stats.iterations))
            if loc:
                fg, fo = float(m.GetFitnessScore(2.0)), float(o.GetFitnessScore(2.0))
                assert (fg == fo) or (np.isnan(fg) and np.isnan(fo)) or abs(fg - fo) <= 1e-6 * max(abs(fo), 1e-30), (name, k, fg, fo)
                rec["fitness"] = fo
            hist.append(rec)
            Tprev = T_ref
        return hist
    finally:
        m.close()
        o.close()


DEGENERATE = ("deg_ivox_empty", "deg_ivox_tiny", "deg_ivox_far", "deg_icp_tiny12", "deg_icp_far", "deg_ndt_empty", "deg_ndt_tiny", "deg_ndt_far",
              "deg_loam_nocorner", "deg_loam_tiny", "deg_loam_far", "deg_kd_tiny", "deg_kd_far")  # = tests/test_ref_pin.py::DEGENERATE
