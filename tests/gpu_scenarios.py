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


def run_scenario(name: str, sc: dict = None, tie_break_by_id: bool = False) -> list:
    """Drive the HIP matcher and the oracle through one scenario; raises AssertionError on the first difference.  Returns the per-frame history.
    tie_break_by_id: the oracle orders candidates at EXACTLY equal distance by insertion id (the device's (d2, id) keys) instead of leaving them to
    libstdc++'s introselect like the reference -- only to prove that a difference is a distance tie and nothing else (oracle/flo_api.h)."""
    from oracle import oracle as O
    O.set_tie_break_by_id(tie_break_by_id)
    try:
        return _run_scenario(name, sc)
    finally:
        O.set_tie_break_by_id(False)


def _run_scenario(name: str, sc: dict = None) -> list:
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

def explain_by_ties(name: str) -> dict:
    """For a scenario on which run_scenario raises: find the first (frame, iteration) at which the HIP path and the oracle part ways, cut both Matches right
    after that iteration (optimization_iter_num = iteration + 1, so the correspondences both sides report ARE that iteration's) and check that the
    divergence is an exact distance tie and nothing else:
      * every row whose neighbour list differs carries the oracle's tie flag (the search met equal float distances across the 5th / 6th candidate or
        between the nearest two), and there is at least one such row;
      * for each of those rows the device's five neighbours and the oracle's five have THE SAME five float squared distances to the query (computed here
        from the map and the pose the iteration started from, in the kernels' association order): one equidistant point stands in for another;
      * counts, valid flags, n_valid of every iteration up to and including that one are equal.
    The reference leaves candidates at equal distance in libstdc++'s introselect order (ivox_map.cpp:24-36: std::nth_element on distances); the device's
    keys are (d2, map image slot).  PointToPlane_IVOX only (the kind whose oracle keeps tie flags)."""
    sc = refpin.make_scenario(name)
    mode, y, loc = sc["mode"], sc["y"], bool(sc.get("loc", False))
    # (localization mode only: the iteration budget is a constructor argument, and in mapping mode a shorter budget changes the earlier frames' poses and
    # with them the map the frame in question sees.  Mapping-mode scenarios: run_scenario(..., tie_break_by_id=True).)
    assert mode == "PointToPlane_IVOX" and loc and "ivox_capacity" not in sc

    def drive(yy, n_frames):
        m = reg.make_matcher(mode, yy, is_localization_mode=loc)
        o = util.oracle_for(mode, yy, loc)
        m.AddCloudToLocalMap(sc["init_clouds"]); o.AddCloudToLocalMap(*sc["init_clouds"])
        Tprev, out = np.eye(4), None
        for k, f in enumerate(sc["frames"][:n_frames]):
            guess = f["absolute_guess"] if "absolute_guess" in f else Tprev @ f["guess_step"]
            T = np.array(guess, dtype=np.float64).copy()
            m.Match(util.cluster_for(mode, f["scan"], None), T, update_map=True)
            _, T_ref = o.Match(f["scan"], np.array(guess, dtype=np.float64), update_map=True)
            out = dict(frame=k, guess=np.array(guess), glog=m.iteration_log(), olog=o.iteration_log(), gcor=m.correspondences(0), ocor=o.correspondences(0),
                       tie=o.tie_rows(), ties=int(o.counters().tie_queries), map=o.map_dump(0), scan=f["scan"])
            yield out
            Tprev = T_ref
        m.close(); o.close()

    first = None
    for st in drive(y, len(sc["frames"])):
        (Tg, nvg, srg), (To, nvo, sro) = st["glog"], st["olog"]
        assert np.array_equal(nvg, nvo), (name, st["frame"])
        for i in range(len(nvo)):
            if abs(srg[i] - sro[i]) > 1e-9 * max(1.0, abs(sro[i])) or not np.allclose(Tg[i], To[i], rtol=0, atol=1e-12):
                first = (st["frame"], i)
                break
        if first:
            assert st["ties"] >= 1, (name, first, "the oracle met no distance tie in this Match")
            break
    assert first is not None, f"{name}: no divergence found"
    k, i = first
    cut = None
    for st in drive(dict(y, optimization_iter_num=i + 1), k + 1):
        cut = st
    (Tg, nvg, srg), (To, nvo, sro) = cut["glog"], cut["olog"]
    assert len(nvo) == i + 1 and np.array_equal(nvg, nvo)
    for j in range(i):  # everything before the divergence is the usual exact agreement
        assert abs(srg[j] - sro[j]) <= 1e-9 * max(1.0, abs(sro[j])) and np.allclose(Tg[j], To[j], rtol=0, atol=1e-12), (name, j)
    (ids, cnt, valid), (ids_r, cnt_r, valid_r) = cut["gcor"], cut["ocor"]
    assert np.array_equal(cnt, cnt_r) and np.array_equal(valid, valid_r)
    bad = (ids[:, 0] != ids_r[:, 0]) | (np.sort(ids, 1) != np.sort(ids_r, 1)).any(1)
    tie = cut["tie"]
    assert bad.any() and tie is not None and not (bad & ~tie).any(), (name, int(bad.sum()), np.flatnonzero(bad & ~tie)[:4])
    Tq = To[i - 1] if i >= 1 else cut["guess"]  # the pose iteration i searched with (identical on both sides: checked above)
    mp = cut["map"].astype(np.float32)
    rows = []
    for r in np.flatnonzero(bad):
        p = cut["scan"][r].astype(np.float64)
        q = (Tq[:3, :3] @ p + Tq[:3, 3]).astype(np.float32)

        def d2(ix):
            d = q - mp[ix]
            return np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        dg = sorted(float(d2(ix)) for ix in ids[r] if ix >= 0)
        do = sorted(float(d2(ix)) for ix in ids_r[r] if ix >= 0)
        assert dg == do, (name, int(r), dg, do)  # the same five distances: an equidistant point took another's place
        rows.append(dict(row=int(r), gpu=[int(v) for v in ids[r]], oracle=[int(v) for v in ids_r[r]], d2=dg))
    return dict(frame=k, iteration=i, rows=rows, d_sum_res=float(srg[i] - sro[i]))


# Scenarios of the round-6 GPU runs (540 in all) in which ONE query of ONE middle iteration meets an exact float distance tie across the 5th / 6th
# candidate (the oracle counts it: counters().tie_queries == 1): the reference's choice is libstdc++'s introselect permutation, the device takes the lower
# map index.  That iteration's residual sum differs by ~1e-5 relative, its pose by 4e-6 m; the iterations after it and the final rows agree again.
TIE_SCENARIOS = ("fuzz319", "lfuzz58")


def explain_difference_as_tie(name: str):
    """A scenario on which run_scenario raises is 'a distance tie and nothing else' if EITHER the whole scenario is equal again once the oracle orders
    exactly tied candidates by insertion id (run_scenario(tie_break_by_id=True): the device's (d2, image slot) keys agree with that order whenever the lower
    id also sits in the lower slot) OR, in localization mode, explain_by_ties holds at the first differing iteration.  Returns a description, or None."""
    try:
        run_scenario(name, tie_break_by_id=True)
        return dict(how="equal with the oracle's tie-by-id switch")
    except AssertionError:
        pass
    try:
        return dict(how="explain_by_ties", **explain_by_ties(name))
    except AssertionError:
        return None

