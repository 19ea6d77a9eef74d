"""one scenario of tests/refpin.py on the GPU, frame by frame, with everything printed where the HIP path and the oracle differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import refpin, util
name = sys.argv[1]
sc = refpin.make_scenario(name)
mode, y, loc = sc["mode"], sc["y"], bool(sc.get("loc", False))
cap = sc.get("ivox_capacity")
if cap is not None:
    os.environ["FLS_IVOX_CAPACITY"] = str(cap)
m = reg.make_matcher(mode, y, is_localization_mode=loc) if mode != "LoamFull_KdTree" else reg.make_matcher(mode, y)
o = util.oracle_for(mode, y, loc)
if cap is not None:
    o.set_ivox_capacity(cap)
print(name, mode, y, "capacity", cap)
m.AddCloudToLocalMap(sc["init_clouds"]); o.AddCloudToLocalMap(*sc["init_clouds"])
Tprev = np.eye(4)
for k, f in enumerate(sc["frames"]):
    guess = f["absolute_guess"] if "absolute_guess" in f else Tprev @ f["guess_step"]
    T = np.array(guess, dtype=np.float64).copy()
    ok = m.Match(util.cluster_for(mode, f["scan"], f["corner"]), T, update_map=True)
    ok_ref, T_ref = o.Match(f["scan"], np.array(guess, dtype=np.float64), src1=f["corner"], update_map=True)
    Tg, nvg, srg = m.iteration_log(); To, nvo, sro = o.iteration_log()
    ids, cnt, valid = m.correspondences(0); ids_r, cnt_r, valid_r = o.correspondences(0)
    tie = o.tie_rows()
    bad = (ids[:, 0] != ids_r[:, 0]) | (np.sort(ids, 1) != np.sort(ids_r, 1)).any(1)
    print(f"frame {k}: ok {ok}/{ok_ref} iters {m.stats.iterations}/{o.stats.iterations} n_valid {list(nvg)} / {list(nvo)}")
    print("   sum_res gpu", [float(f'{v:.9g}') for v in srg]); print("   sum_res ora", [float(f'{v:.9g}') for v in sro])
    print("   pose dt/dr per iteration", [tuple(float(f'{v:.2e}') for v in synth.pose_error(Tg[i], To[i])) for i in range(len(nvo))])
    print("   final rows differing", int(bad.sum()), "of them tie-flagged", int((bad & tie).sum()) if tie is not None else None, "tie rows total", int(tie.sum()) if tie is not None else None,
          "tie_queries", int(o.counters().tie_queries), "cnt equal", bool(np.array_equal(cnt, cnt_r)), "valid equal", bool(np.array_equal(valid, valid_r)), "map", m.map_size(0), o.map_size(0))
    for r in np.flatnonzero(bad)[:6]:
        print("     row", r, "gpu", ids[r], "ora", ids_r[r], "tie", bool(tie[r]) if tie is not None else None, "valid", valid[r], valid_r[r])
    Tprev = T_ref
