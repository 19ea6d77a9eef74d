"""profiles/traffic_ivox_knn.json from the raw rocprofv3 --pmc CSVs of tools/prof_round4.sh (one counter group per pass).
usage: python tools/make_traffic_json.py <dir with pmc1..pmc5 counter_collection csv copies> <trace avg launch us> > profiles/traffic_ivox_knn.json
Units as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KB per dispatch, FETCH_SIZE doubled on gfx950."""
import collections, csv, glob, json, os, sys

d, launch_us = sys.argv[1], float(sys.argv[2])
vals = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ivox_knn_kernel<4, false" not in k and "ivox_knn_kernelILi4ELb0" not in k:
            continue  # (the counting variant <4, true, ...> runs once, outside the timed region)
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))


def active_mean(name):
    v = vals.get(name, [])
    if not v:
        return None
    top = max(v)
    a = [x for x in v if x > 0.5 * top]  # early-exit launches (converged Match) are tiny
    return sum(a) / len(a)


fetch, write = active_mean("FETCH_SIZE"), active_mean("WRITE_SIZE")
out = {"kernel": "ivox_knn_kernel<4,false,true,*,true>", "source": "rocprofv3 --kernel-trace --pmc <one group per pass>, python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-extras (tools/prof_round4.sh, profiles/r04_d_pmc_summary.txt)",
       "fetch_size_kb_per_launch_raw": fetch, "write_size_kb_per_launch_raw": write, "fetch_correction": 2.0,
       "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None}
rd = active_mean("TCP_TCC_READ_REQ_sum")
hit, miss = active_mean("TCC_HIT_sum"), active_mean("TCC_MISS_sum")
if rd is not None:
    out["l2_read_requests_per_launch"] = rd
    out["l2_read_bytes_per_launch"] = rd * 128.0
if hit is not None and miss is not None and hit + miss > 0:
    out["l2_hit_rate"] = hit / (hit + miss)
valu = active_mean("SQ_INSTS_VALU")
if valu is not None:
    floor_us = valu * 4.0 / 1024.0 / 2.4e3
    out["valu_wave_instructions_per_launch"] = valu
    out["instruction_floor_us"] = floor_us
    out["valu_busy_pct"] = 100.0 * floor_us / launch_us
wc, wa, wi, ai = active_mean("SQ_WAVE_CYCLES"), active_mean("SQ_WAIT_ANY"), active_mean("SQ_WAIT_INST_ANY"), active_mean("SQ_ACTIVE_INST_ANY")
if wc:
    if wa is not None: out["wave_wait_pct"] = 100.0 * wa / wc
    if wi is not None: out["wave_issue_stall_pct"] = 100.0 * wi / wc
    if ai is not None: out["wave_issuing_pct"] = 100.0 * ai / wc
out["trace_avg_launch_us"] = launch_us
out["note"] = ("mean over the active launches of the first-iteration and later-iteration instantiations; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, "
               "WRITE_SIZE uncorrected; l2_read_bytes assumes 128-B TCP->TCC read requests; instruction_floor = SQ_INSTS_VALU wave-instructions x 4 cycles / 1024 SIMDs at "
               "the nominal 2.4 GHz; valu_busy = instruction_floor / the trace's launch duration; wave_* as fractions of SQ_WAVE_CYCLES.  Round 3: the neighbour lists leave "
               "the kernel as 20-byte id rows (32-byte stride), not as 80-byte gathered rows.  Round 4: the voxel look-up goes through the two-level brick image "
               "(one directory load per query + slab offsets) instead of the bounding-box window: same traffic, same instruction count")
print(json.dumps(out, indent=1))
