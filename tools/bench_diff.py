"""Regression guard between rounds (VERDICT r5 weak #12 / next #1b): compare a new bench line with the previous round's record LEG BY LEG and
print every leg that moved by more than a threshold (default 10 %).

    python tools/bench_diff.py NEW [OLD] [--threshold 0.10] [--all]

NEW / OLD: a file holding one bench.py JSON line (profiles/rNN_*_bench_full.json, gpurun_out/.../bench.json) or a driver record (BENCH_rNN.json:
`parsed` + the last 2,000 characters of the line in `tail`).  OLD defaults to the newest BENCH_r*.json at the repository root, completed -- for the
legs its truncated tail no longer holds -- from the newest builder-kept full line of that round under profiles/ (named in the output).
The legs are bench.py's LEGS table: since round 6 the line ends with them (`legs`, flat), so the driver's tail always carries all of them; lines of
earlier rounds are reduced with the same table.  Exit status 1 when a leg got WORSE by more than the threshold (better / unchanged: 0)."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # LEGS, legs_from_line (no GPU needed to import)


def legs_of_file(path):
    """-> (legs, description).  Accepts a bench line, a driver record, or a file whose last JSON line is a bench line."""
    with open(path) as f:
        txt = f.read()
    try:
        d = json.loads(txt)
    except ValueError:
        d = None
        for ln in reversed(txt.strip().splitlines()):  # a log: the last line that parses
            try:
                d = json.loads(ln)
                break
            except ValueError:
                continue
        if d is None:
            raise SystemExit(f"{path}: no JSON line found")
    if "parsed" in d and "tail" in d:  # driver record
        legs = bench.legs_from_line(d.get("parsed") or {})
        tail = d.get("tail") or ""
        m = re.search(r'"legs":\s*(\{[^{}]*\})', tail)
        if m:
            legs.update(json.loads(m.group(1)))
            return legs, f"{os.path.basename(path)} (driver record, legs from its tail)"
        # rounds 1-5: recover what flat objects the truncated tail still holds
        for key in ("inclusive_h2d", "cpu_baseline", "cpu_baseline_ref", "loop_closure"):
            mm = re.search(r'"%s":\s*(\{[^{}]*\})' % key, tail)
            if mm:
                try:
                    legs.update(bench.legs_from_line({key: json.loads(mm.group(1))}))
                except ValueError:
                    pass
        rnd = re.search(r"BENCH_r(\d+)", os.path.basename(path))
        note = f"{os.path.basename(path)} (driver record, {len(legs)} legs recoverable)"
        if rnd:
            full = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{int(rnd.group(1)):02d}_*bench_full*.json")))  # (by name: a checkout has no mtimes)
            full = [p for p in full if "slow_box" not in p and "other_box" not in p]
            if full:
                with open(full[-1]) as f:
                    more = bench.legs_from_line(json.loads(f.read().strip().splitlines()[-1]))
                n0 = len(legs)
                for k, v in more.items():
                    legs.setdefault(k, v)
                note += f" + {len(legs) - n0} legs from profiles/{os.path.basename(full[-1])} (builder-kept line of the same round)"
        return legs, note
    legs = dict(d.get("legs") or {})
    for k, v in bench.legs_from_line(d).items():
        legs.setdefault(k, v)
    return legs, os.path.basename(path)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    thr = 0.10
    if "--threshold" in sys.argv:
        thr = float(sys.argv[sys.argv.index("--threshold") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--threshold") + 1]]
    show_all = "--all" in sys.argv
    if not args:
        raise SystemExit(__doc__)
    new_path = args[0]
    if len(args) > 1:
        old_path = args[1]
    else:
        recs = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")))
        if not recs:
            raise SystemExit("no BENCH_r*.json to compare with")
        old_path = recs[-1]
    new, new_desc = legs_of_file(new_path)
    old, old_desc = legs_of_file(old_path)
    print(f"bench_diff: NEW = {new_desc}\n            OLD = {old_desc}\n            threshold {100 * thr:.0f} %")
    worse = better = same = 0
    rows = []
    for name, (_, sense, _) in bench.LEGS.items():
        if name not in new or name not in old or old[name] == 0:
            if name in new or name in old:
                rows.append((name, old.get(name), new.get(name), None, "only in " + ("NEW" if name in new else "OLD")))
            continue
        rel = new[name] / old[name] - 1.0
        gain = -rel if sense == "lower" else rel  # > 0: better
        tag = "same"
        if gain < -thr:
            tag = "WORSE"; worse += 1
        elif gain > thr:
            tag = "better"; better += 1
        else:
            same += 1
        rows.append((name, old[name], new[name], rel, tag))
    for name, o, n, rel, tag in rows:
        if show_all or tag not in ("same",):
            so = "-" if o is None else f"{o:.5g}"
            sn = "-" if n is None else f"{n:.5g}"
            sr = "" if rel is None else f"{100 * rel:+.1f} %"
            print(f"  {name:20s} {so:>10s} -> {sn:>10s}  {sr:>9s}  {tag}")
    print(f"bench_diff: {worse} legs worse, {better} better, {same} within {100 * thr:.0f} % ({len(bench.LEGS)} legs known)")
    return 1 if worse else 0


if __name__ == "__main__":
    sys.exit(main())
