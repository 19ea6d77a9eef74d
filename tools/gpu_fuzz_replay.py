"""HIP path vs the CPU oracle on the scenarios the oracle was pinned on against the compiled reference (tests/refpin.py), on a GPU box:
    python tools/gpu_fuzz_replay.py mapping [first [count]]   seeded random mapping-mode replays `fuzz<seed>` (kind by seed % 4; ~1-2 s each)
    python tools/gpu_fuzz_replay.py long    [first [count]]   `fuzzL<seed>`: 8-14 frames each
    python tools/gpu_fuzz_replay.py loc     [first [count]]   localization mode + GetFitnessScore, `lfuzz<seed>`
    python tools/gpu_fuzz_replay.py deg                       the 13 degenerate scenarios (+ the ICP <= 10 points abort)
Per frame: the assertions of tests/gpu_scenarios.py::run_scenario.  One line per scenario, `failed:` list at the end, exit status 1 if any failed.
(`python tools/gpu_fuzz_replay.py <first> <count>` = `mapping <first> <count>`, the round-5 form.)"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_scenarios

args = sys.argv[1:]
what = "mapping"
if args and not args[0].lstrip("-").isdigit():
    what = args.pop(0)
first = int(args[0]) if len(args) > 0 else 0
count = int(args[1]) if len(args) > 1 else 24
if what == "deg":
    names = list(gpu_scenarios.DEGENERATE) + ["deg_icp_le10"]
else:
    prefix = {"mapping": "fuzz", "long": "fuzzL", "loc": "lfuzz"}[what]
    names = [f"{prefix}{s}" for s in range(first, first + count)]
bad, ties = [], []
t_all = time.perf_counter()
for name in names:
    t0 = time.perf_counter()
    try:
        sc = gpu_scenarios.refpin.make_scenario(name)
        hist = gpu_scenarios.run_scenario(name, sc)
        print(name, sc["mode"], "OK", "ok=", [h["ok"] if h["ok"] is None else int(h["ok"]) for h in hist], "iters=", [h["iters"] for h in hist], "upd=", [h["upd"] for h in hist],
              ("fitness= " + " ".join(f"{h['fitness']:.6g}" for h in hist if "fitness" in h)) if any("fitness" in h for h in hist) else "",
              f"{time.perf_counter() - t0:.1f}s", flush=True)
    except BaseException as e:  # (pytest.raises failures derive from BaseException)
        if isinstance(e, KeyboardInterrupt):
            raise
        # a difference that disappears when the oracle orders exactly tied kNN candidates by insertion id (the device's keys) is a distance tie: reported, not failed
        tie_only = None
        if isinstance(e, AssertionError):
            try:
                tie_only = gpu_scenarios.explain_difference_as_tie(name)
            except BaseException:
                tie_only = None
        if tie_only:
            ties.append(name)
            print(name, "TIE: differs by an exact distance tie and nothing else (tests/gpu_scenarios.py::explain_difference_as_tie):", tie_only, flush=True)
        else:
            bad.append(name)
            print(name, "FAIL", repr(e)[:400], flush=True)
            traceback.print_exc(limit=3)
print(f"{what}: {len(names) - len(bad) - len(ties)} of {len(names)} scenarios equal to the oracle in {time.perf_counter() - t_all:.0f} s; "
      f"equal up to an exact distance tie (explain_difference_as_tie): {ties}; failed: {bad}", flush=True)
sys.exit(1 if bad else 0)
