"""MatchBatch (lanes = 3, then 1, then 4) of four jobs on a fresh handle, N times; every pose against a fresh handle's pose (bitwise).
usage: python tools/dbg_batch_stress.py N [icp|ndt|ivox] [scale]   (ndt at scale 1.0: 115,200-point scans -- the exact sort's pre-enqueued levels and the fused
Gauss-Newton tail of several lanes at once)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util
N = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "icp"
mode, y, cid, scale, loc = {"icp": ("IcpOptimized", reg.YAML_NCLT_ICP, 0, 1.0, True), "ndt": ("IncrementalNDT", reg.YAML_NCLT_NDT, 2, 0.05, False),
                              "ivox": ("PointToPlane_IVOX", reg.YAML_NCLT_IVOX, 1, 0.2, False)}[kind]
if len(sys.argv) > 3: scale = float(sys.argv[3])
cfgs = [synth.make_config(cid, job=j, scale=scale) for j in range(4)]
maps = [cfgs[0]["map"]]
clusters = [util.cluster_for(mode, c["scan"], None) for c in cfgs]
good = []
for j in range(4):
    f = reg.make_matcher(mode, y, is_localization_mode=loc); f.AddCloudToLocalMap(maps)
    T = np.eye(4); f.Match(clusters[j], T, update_map=False); good.append((T.copy(), f.stats.iterations, f.stats.n_valid, f.stats.n_source)); f.close()
bad = collections.Counter()
for rep in range(N):
    m = reg.make_matcher(mode, y, is_localization_mode=loc); m.AddCloudToLocalMap(maps)
    for lanes in (3, 1, 4):
        oks, Ts, st = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=lanes)
        for j in range(4):
            if not np.array_equal(Ts[j], good[j][0]):
                bad[(lanes, j)] += 1
                print("rep", rep, "lanes", lanes, "job", j, "it", st[j].iterations, "nv", st[j].n_valid, "src", st[j].n_source, "good", good[j][1:], "dT", float(np.abs(Ts[j] - good[j][0]).max()), flush=True)
    m.close()
print(kind, "reps", N, "mismatches", dict(bad), flush=True)
