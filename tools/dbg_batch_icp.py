import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from tests import util
mode, y, cid, scale, loc = "IcpOptimized", reg.YAML_NCLT_ICP, 0, 1.0, True
cfgs = [synth.make_config(cid, job=j, scale=scale) for j in range(4)]
maps = [cfgs[0]["map"]]
clusters = [util.cluster_for(mode, c["scan"], None) for c in cfgs]
for rep in range(3):
    m = reg.make_matcher(mode, y, is_localization_mode=loc); m.AddCloudToLocalMap(maps)
    oks, Ts, stats = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=3)
    oks1, Ts1, stats1 = m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=1)
    for j in range(4):
        f = reg.make_matcher(mode, y, is_localization_mode=loc); f.AddCloudToLocalMap(maps)
        T = np.eye(4); ok = f.Match(clusters[j], T, update_map=False)
        T2 = np.eye(4); ok2 = f.Match(clusters[j], T2, update_map=False)
        print(rep, j, "fresh it", f.stats.iterations, "nv", f.stats.n_valid, "src", f.stats.n_source, "| lanes3 it", stats[j].iterations, stats[j].n_valid, stats[j].n_source, "| lanes1 it", stats1[j].iterations, stats1[j].n_valid, stats1[j].n_source,
              "| fresh==l3", np.array_equal(T, Ts[j]), "fresh==l1", np.array_equal(T, Ts1[j]), "l3==l1", np.array_equal(Ts[j], Ts1[j]), "fresh==fresh2", np.array_equal(T, T2), flush=True)
        f.close()
    m.close()
