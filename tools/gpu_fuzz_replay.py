"""HIP path vs the CPU oracle on the seeded random mapping-mode scenarios of tests/refpin.py::make_fuzz_scenario -- the scenarios on which the
oracle was checked against the compiled reference (profiles/r05_ref_pin_fuzz_200_scenarios.log), with the assertions of
tests/test_gpu_mapping_replay.py::run_replay after every frame (return value, iterations, per-iteration n_valid / pose, flags, counts, ids, map sizes).
usage: python tools/gpu_fuzz_replay.py [first_seed [n_seeds]]        (default 0 24; ~1-2 s per scenario)
STATUS: written at the very end of round 5; the last seconds of the round's GPU budget ran seeds 0-3 (one per kind): all four equal the oracle
through every frame (profiles/r05_late_gpu_fuzz_replay_first_seeds.log).  Seeds from 4 on have not been run on a GPU."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import refpin, util
from tests.test_gpu_mapping_replay import run_replay

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = []
for seed in range(first, first + count):
    sc = refpin.make_fuzz_scenario(seed)
    t0 = time.perf_counter()
    try:
        if "ivox_capacity" in sc:
            # the LRU capacity is a constructor constant of the reference (ivox_map.h); the handle takes it through its test hook
            os.environ["FLS_IVOX_CAPACITY"] = str(sc["ivox_capacity"])
            orig = util.oracle_for
            def with_cap(mode, y, loc=False, _cap=sc["ivox_capacity"]):
                o = orig(mode, y, loc); o.set_ivox_capacity(_cap); return o
            util.oracle_for = with_cap
        r, hist = run_replay(sc["name"], r=sc)
        print(seed, sc["mode"], "OK", "ok=", [int(h["ok"]) for h in hist], "iters=", [h["iters"] for h in hist], "upd=", [h["upd"] for h in hist],
              f"{time.perf_counter() - t0:.1f}s", flush=True)
    except Exception as e:
        bad.append(seed)
        print(seed, sc["mode"], "FAIL", repr(e)[:400], flush=True)
        traceback.print_exc(limit=2)
    finally:
        if "ivox_capacity" in sc:
            os.environ.pop("FLS_IVOX_CAPACITY", None)
            util.oracle_for = orig
print("failed seeds:", bad)
sys.exit(1 if bad else 0)
