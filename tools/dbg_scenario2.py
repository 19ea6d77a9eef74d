"""lfuzz58 cut after the iteration that differs: which query, which neighbours"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import refpin, util
from oracle import oracle as O
name, cut, tie = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
O.set_tie_break_by_id(bool(tie))
sc = refpin.make_scenario(name)
mode, loc = sc["mode"], bool(sc.get("loc", False))
y = dict(sc["y"], optimization_iter_num=cut)
m = reg.make_matcher(mode, y, is_localization_mode=loc); o = util.oracle_for(mode, y, loc)
m.AddCloudToLocalMap(sc["init_clouds"]); o.AddCloudToLocalMap(*sc["init_clouds"])
mp = o.map_dump(0)
Tprev = np.eye(4)
for k, f in enumerate(sc["frames"][:int(sys.argv[4]) if len(sys.argv) > 4 else 2]):
    guess = f["absolute_guess"] if "absolute_guess" in f else Tprev @ f["guess_step"]
    T = np.array(guess, dtype=np.float64).copy()
    ok = m.Match(util.cluster_for(mode, f["scan"], f["corner"]), T, update_map=True)
    ok_ref, T_ref = o.Match(f["scan"], np.array(guess, dtype=np.float64), src1=f["corner"], update_map=True)
    Tg, nvg, srg = m.iteration_log(); To, nvo, sro = o.iteration_log()
    ids, cnt, valid = m.correspondences(0); ids_r, cnt_r, valid_r = o.correspondences(0)
    tieflag = o.tie_rows()
    bad = (ids != ids_r).any(1)  # fully ordered
    print(f"frame {k}: iters {m.stats.iterations}/{o.stats.iterations} n_valid {list(map(int,nvg))} / {list(map(int,nvo))} sum_res last {srg[-1]:.9f} / {sro[-1]:.9f} rows differing {int(bad.sum())} tie rows {int(tieflag.sum()) if tieflag is not None else None} valid differs {int((valid != valid_r).sum())} cnt differs {int((cnt != cnt_r).sum())}")
    # the pose BEFORE the last iteration = log entry cut-2 (poses are logged after each update)
    if len(To) >= 2:
        Tq = To[-2]
    else:
        Tq = guess
    for r in list(np.flatnonzero(bad)[:4]) + list(np.flatnonzero(tieflag)[:4] if tieflag is not None else []):
        p = f["scan"][r].astype(np.float64)
        q = (Tq[:3, :3] @ p + Tq[:3, 3]).astype(np.float32)
        def d2(i):
            w = mp[i].astype(np.float32); d = q - w
            return np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        print("   row", int(r), "tie", bool(tieflag[r]) if tieflag is not None else None, "valid", int(valid[r]), int(valid_r[r]))
        print("      gpu ids", ids[r], [float(d2(i)) for i in ids[r] if i >= 0])
        print("      ora ids", ids_r[r], [float(d2(i)) for i in ids_r[r] if i >= 0])
    Tprev = T_ref
