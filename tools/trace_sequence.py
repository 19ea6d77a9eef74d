"""Chronological view of a rocprofv3 kernel-trace CSV (+ optional memory-copy CSV): every dispatch of a time window with its start offset,
duration and the idle gap in front of it.  usage: trace_sequence.py kernel_trace.csv [memory_copy_trace.csv] [--skip FRAC] [--n N]
Prints N consecutive records starting FRAC of the way into the run (default: the middle), so that one steady-state call can be read
launch by launch."""
import csv, sys
opt = {sys.argv[i]: sys.argv[i + 1] for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
frac, count = float(opt.get("--skip", 0.5)), int(opt.get("--n", 80))
rows = []
for r in csv.DictReader(open(args[0])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fls::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[-60:]))
if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "[copy %s %s B]" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
rows.sort()
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
prev = None
for s, e, n in rows[i0:i0 + count]:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:7.2f}  {n}")
    prev = e
