"""The regression guard between rounds (tools/bench_diff.py, bench.py::LEGS): on the records this repository holds it must name the regression that went
unnoticed in round 5 -- the large-cloud exact VoxelGrid legs between the round-4 and round-5 lines -- and a driver record whose
2,000-character tail ends with the flat `legs` object must be read from that tail alone."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run_diff(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_diff.py"), *args], capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout + p.stderr


def test_round5_line_against_round4_record_flags_the_unreported_regression():
    rc, out = run_diff(os.path.join(ROOT, "profiles", "r05_late_bench_full.json"), os.path.join(ROOT, "BENCH_r04.json"))
    assert rc == 1, out
    lines = {ln.split()[0]: ln for ln in out.splitlines() if ln.startswith("  ")}
    assert "WORSE" in lines["map_loam_kf_ms"] and "WORSE" in lines["map_icp_kf_ms"], out  # 2.28 -> 5.5 ms, 0.89 -> 1.46 ms
    assert "better" in lines["ndt_host_us"] and "better" in lines["icp_host_us"], out       # what round 5 did report


def test_legs_are_flat_short_and_last(tmp_path):
    with open(os.path.join(ROOT, "profiles", "r05_late_bench_full.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    legs = bench.legs_from_line(line)
    assert len(legs) >= 30 and all(isinstance(v, float) for v in legs.values())
    assert len(json.dumps(legs)) < 1500, "the legs must fit into the 2,000 characters of the line the driver keeps"
    # a driver record that holds only `parsed` (head keys) and the TAIL of the line: every leg must come back from the tail
    line["legs"] = legs
    txt = json.dumps(line)
    rec = {"parsed": {k: line[k] for k in ("metric", "value", "ms_per_step")}, "tail": txt[-2000:] + "\n---- stderr ----\nnoise\n"}
    a = tmp_path / "BENCH_r99.json"
    a.write_text(json.dumps(rec))
    b = tmp_path / "new.json"
    worse = dict(line)
    worse["legs"] = dict(legs, map_loam_kf_ms=legs["map_loam_kf_ms"] * 1.5)
    worse["mapping_mode"] = json.loads(json.dumps(line["mapping_mode"]))
    worse["mapping_mode"]["loam_full"]["default_device_filter_exact"]["ms_keyframe_update_only"] *= 1.5
    b.write_text(json.dumps(worse))
    rc, out = run_diff(str(b), str(a))
    assert rc == 1 and "legs from its tail" in out, out
    flagged = [ln.split()[0] for ln in out.splitlines() if "WORSE" in ln]
    assert flagged == ["map_loam_kf_ms"], out
