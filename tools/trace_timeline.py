"""Timeline view of a rocprofv3 kernel trace CSV: per kernel (short name) duration statistics AND the idle gap in front of each launch
(start - previous end on the same queue), so that launch boundaries show up next to kernel times.  Early-exit launches (< thr us) are
listed separately."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    for p in ("void ", "fls::"):
        n = n.replace(p, "")
    return n[-46:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    k = short(r["Kernel_Name"]) + ("" if d > thr else "  [early exit]")
    dur[k].append(d)
    if prev_end is not None and (s - prev_end) / 1e3 < 50.0:  # gaps inside a Match (the host is away between Matches)
        gap[k].append((s - prev_end) / 1e3)
    prev_end = e
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = gap.get(k, [])
    print(f"{k:62s} n {len(v):5d}  dur median {med(v):7.2f} mean {sum(v)/len(v):7.2f} min {min(v):6.2f} max {max(v):7.2f} us | gap before: median {med(g):6.2f} mean {(sum(g)/len(g)) if g else float('nan'):6.2f} us (n {len(g)})")
