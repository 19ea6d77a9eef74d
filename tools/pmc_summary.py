"""Sum rocprofv3 --pmc counter_collection.csv per kernel / counter (average per dispatch over active dispatches)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        top = v[len(v) // 2:]  # the active half (early-exit launches are tiny)
        print(f"    {c:28s} n={len(v):4d} median_active={top[len(top)//2]:14.0f} max={v[-1]:14.0f}")
