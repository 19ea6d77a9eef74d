"""Per-kernel summary of a rocprofv3 kernel trace CSV: count, active (non early-exit) launches, mean/min/max us."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.5
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][-44:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    act = [x for x in v if x > thr]
    print(f"{k:46s} calls {len(v):5d} active {len(act):5d} mean_active {sum(act)/max(len(act),1):8.2f} us  min {min(v):6.2f} max {max(v):7.2f} total {sum(v)/1e3:8.3f} ms")
