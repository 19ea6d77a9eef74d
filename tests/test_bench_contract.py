"""The bench.py output contract (one JSON line), checked on the line committed from the last GPU run of the round
(profiles/r02_h_bench_full.json) and on bench.py's own source (metric / config strings = BASELINE.json's)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_h_bench_full.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    # (lines committed before the metric string was read from BASELINE.json spell the multiplication sign and the arrow in ASCII)
    norm = lambda t: t.replace("\u00d7", " x ").replace("\u2192", "->").replace("  ", " ")
    assert norm(line["metric"]) == norm(base["metric"]) and line["unit"] == "scans/s"
    import bench
    assert bench.baseline_metric() == base["metric"]
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "configs[1]" in line["config"]["workload"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]  # whole-job throughput = steps / timed region
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == line["unit"]
    # the extras the round-1 verdict asked for
    for k in ("configs", "mapping_mode", "inclusive_h2d", "pose_err_vs_oracle", "c5_batch"):
        assert k in line, k
    assert set(line["configs"]) == {"configs[0]", "configs[2]", "configs[3]"}
    assert line["c5_batch"]["jobs"] == 512 and line["c5_batch"]["converged_jobs"] == 512 and line["c5_batch"]["distinct_poses"] == 512
    assert line["pose_err_vs_oracle"]["dt_m"] < 1e-4 and line["pose_err_vs_oracle"]["dR_rad"] < 1e-4 and line["pose_err_vs_oracle"]["same_iterations"]
