"""GPU <-> the reference's own compiled classes, DIRECTLY (no oracle in between), on the seeded random scenarios of tests/refpin.py: both sit behind
std::shared_ptr<RegistrationInterface> in oracle/_ref/gpu_vs_ref (tests/harness/gpu_vs_ref.cpp: the reference's sources compiled verbatim + the product's
adapter compiled against the reference's real headers), one fresh process per scenario, the pipeline's call sequence.  Per frame: same return value, pose
<= 1e-8 m / rad (the contract is 1e-4), fitness equal in localization mode.
    python tools/gpu_vs_ref_fuzz.py mapping|loc|long [first [count]]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import refpin
from tests.test_gpu_vs_ref import EXE, write_scenario, parse_out, params_for

what = sys.argv[1] if len(sys.argv) > 1 else "mapping"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
prefix = {"mapping": "fuzz", "long": "fuzzL", "loc": "lfuzz"}[what]
# scenarios whose Gauss-Newton systems amplify last-bit differences beyond this tool's 1e-8 (still inside the 1e-4 contract); tools/cond_scenario.py prints the numbers
SENSITIVE = {"fuzz355": "frame 0 starts with 13 valid points, cond(H) 3e8 and a step of 4.7 -- the summation order of H alone moves such a solution by up to 1e-7; "
                        "4.5e-12 with the one-lane LDL^T of rounds 2-6, 7.3e-8 with the rows-in-lanes LDL^T: profiles/r06_av_fuzz355_conditioning.txt"}
bad, ties, worst = [], [], 0.0
t_all = time.perf_counter()
for seed in range(first, first + count):
    name = f"{prefix}{seed}"
    sc = refpin.make_scenario(name)
    with tempfile.TemporaryDirectory() as td:
        scn, outp = os.path.join(td, "scn.bin"), os.path.join(td, "out.txt")
        write_scenario(scn, sc, params_for(sc))
        t0 = time.perf_counter()
        run = subprocess.run([EXE, scn, outp], capture_output=True, text=True, timeout=900)
        try:
            frames, summary = parse_out(outp)
            ok = run.returncode == 0 and len(frames) == len(sc["frames"])
            for fr in frames:
                ok = ok and fr["ref_ok"] == fr["hip_ok"] and fr["dt"] <= 1e-8 and fr["dr"] <= 1e-8
                if sc["loc"]:
                    a, b = fr["ref_fitness"], fr["hip_fitness"]
                    ok = ok and abs(a - b) <= 1e-6 * max(1.0, abs(a))
            wd = max([max(fr["dt"], fr["dr"]) for fr in frames] or [0.0])
            worst = max(worst, wd) if ok else worst
            tag = "OK"
            if not ok:
                # an exact distance tie (tests/gpu_scenarios.py::TIE_SCENARIOS) shows here as a pose difference of ~1e-6 in the frames after it: inside the
                # contract (1e-4), outside this tool's 1e-8
                inside = run.returncode in (0, 1) and len(frames) == len(sc["frames"]) and all(fr["ref_ok"] == fr["hip_ok"] and fr["dt"] <= 1e-4 and fr["dr"] <= 1e-4 for fr in frames)
                if inside and name in ("fuzz319", "lfuzz58"):
                    ties.append(name); tag = "TIE (known: one exact distance tie)"
                elif inside and name in SENSITIVE:
                    ties.append(name); tag = "ILL-CONDITIONED (known: " + SENSITIVE[name] + ")"
                else:
                    bad.append(name); tag = "FAIL"
            print(name, sc["mode"], tag, "frames", len(frames), "ok=", [fr["ref_ok"] for fr in frames], f"worst |dT| {wd:.1e}",
                  f"ref {np.median([f['ref_ms'] for f in frames]):.0f} ms / HIP {np.median([f['hip_ms'] for f in frames]):.2f} ms per Match", f"{time.perf_counter() - t0:.1f}s", flush=True)
            if tag == "FAIL":
                print(run.stdout[-600:], run.stderr[-600:], flush=True)
        except Exception as e:
            bad.append(name)
            print(name, "FAIL", repr(e)[:300], run.stdout[-300:], run.stderr[-300:], flush=True)
print(f"{what}: {count - len(bad) - len(ties)} of {count} scenarios: HIP adapter == compiled reference through RegistrationInterface (worst |dT| {worst:.1e}); known ties / ill-conditioned: {ties}; failed: {bad}; "
      f"{time.perf_counter() - t_all:.0f} s", flush=True)
sys.exit(1 if bad else 0)
