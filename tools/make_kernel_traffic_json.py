"""profiles/traffic_<name>.json for ANY kernel from the raw rocprofv3 --pmc CSVs of tools/prof_round5.sh (one counter group per pass) and the
kernel trace of the same workload.  The round-3/4 generator (make_traffic_json.py) knows one kernel; this one takes the name.
usage: python tools/make_kernel_traffic_json.py <dir with pmc*_counter_collection.csv + kernel_trace.csv> <kernel-name substring> <out name> "<workload>"
Units as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KB per dispatch, FETCH_SIZE doubled on gfx950."""
import collections, csv, glob, json, os, sys

d, sub, name, workload = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
vals = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = []
tr = os.path.join(d, "kernel_trace.csv")
if os.path.exists(tr):
    for r in csv.DictReader(open(tr)):
        if sub in r["Kernel_Name"]:
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def active_mean(v):
    if not v:
        return None
    w = sorted(v)
    top = w[min(len(w) - 1, (9 * len(w)) // 10)]  # the 90th percentile, not the maximum: one slow launch must not define "active"
    a = [x for x in v if x > 0.5 * top]  # early-exit launches (a converged Match, an empty level) are tiny
    return sum(a) / len(a)


launch_us = active_mean(durs)
fetch, write = active_mean(vals.get("FETCH_SIZE")), active_mean(vals.get("WRITE_SIZE"))
out = {"kernel": sub, "workload": workload, "source": "rocprofv3 --kernel-trace --pmc <one group per pass> (tools/prof_round5.sh) -> tools/make_kernel_traffic_json.py, directory " + os.path.basename(os.path.normpath(d)), "launches_in_trace": len(durs), "trace_avg_launch_us": launch_us,
       "fetch_size_kb_per_launch_raw": fetch, "write_size_kb_per_launch_raw": write, "fetch_correction": 2.0,
       "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None}
if out["hbm_bytes_per_launch"] and launch_us:
    out["measured_hbm_GBs"] = out["hbm_bytes_per_launch"] / (launch_us * 1e-6) / 1e9
rd = active_mean(vals.get("TCP_TCC_READ_REQ_sum"))
hit, miss = active_mean(vals.get("TCC_HIT_sum")), active_mean(vals.get("TCC_MISS_sum"))
if rd is not None:
    out["l2_read_bytes_per_launch"] = rd * 128.0
if hit is not None and miss is not None and hit + miss > 0:
    out["l2_hit_rate"] = hit / (hit + miss)
valu = active_mean(vals.get("SQ_INSTS_VALU"))
if valu is not None and launch_us:
    floor_us = valu * 4.0 / 1024.0 / 2.4e3
    out["valu_wave_instructions_per_launch"] = valu
    out["instruction_floor_us"] = floor_us
    out["valu_busy_pct"] = 100.0 * floor_us / launch_us
wc, wa, wi, ai = (active_mean(vals.get(k)) for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
waves = active_mean(vals.get("SQ_WAVES"))
if waves is not None:
    out["waves_per_launch"] = waves
if wc:
    if wa is not None: out["wave_wait_pct"] = 100.0 * wa / wc
    if wi is not None: out["wave_issue_stall_pct"] = 100.0 * wi / wc
    if ai is not None: out["wave_issuing_pct"] = 100.0 * ai / wc
out["note"] = ("mean over the active launches (> half of the 90th percentile); FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE uncorrected; "
               "l2_read_bytes assumes 128-B TCP->TCC read requests; instruction_floor = SQ_INSTS_VALU wave-instructions x 4 cycles / 1024 SIMDs at 2.4 GHz")
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_%s.json" % name), "w") as f:
    json.dump(out, f, indent=1)
print(name, json.dumps({k: out[k] for k in ("trace_avg_launch_us", "hbm_bytes_per_launch", "measured_hbm_GBs", "valu_busy_pct", "wave_wait_pct", "l2_hit_rate") if k in out}))
