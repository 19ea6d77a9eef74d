import sys, numpy as np
sys.path.insert(0, '/root/repo')
from funny_lidar_slam_amd import synth, registration as reg
from tests import util
from scipy.spatial import cKDTree
cfg = synth.make_config(1)
o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX); o.AddCloudToLocalMap(cfg["map"])
ok, T = o.Match(cfg["scan"], np.eye(4), update_map=False)
mp = np.asarray(cfg["map"], np.float32)[:, :3]; sc = np.asarray(cfg["scan"], np.float32)[:, :3]
q = (sc @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
res = 0.5
key = np.round(mp / res).astype(np.int64)
from collections import defaultdict
vox = defaultdict(list)
for i, k in enumerate(map(tuple, key)): vox[k].append(i)
offs = [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if abs(x) + abs(y) + abs(z) <= 2]
rng = np.random.default_rng(0)
idx = rng.choice(len(q), 4000, replace=False)
tot = kept = 0; pv = kv = 0; trips_full = []; trips_pr = []
for i in idx:
    p = q[i]; k = np.round(p / res).astype(np.int64)
    cands = []
    for o3 in offs:
        kk = (k[0] + o3[0], k[1] + o3[1], k[2] + o3[2])
        ids = vox.get(kk)
        if ids: cands.append((kk, ids))
    allids = [j for _, ids in cands for j in ids]
    if len(allids) < 5: continue
    d2 = ((mp[allids] - p) ** 2).sum(1); B = np.sort(d2)[4]
    n_all = len(allids); n_keep = 0
    for kk, ids in cands:
        c = np.array(kk) * res
        e = np.maximum(np.abs(p - c) - 0.251, 0); pv += 1
        if (e * e).sum() * 0.9999 <= B * 1.00001: n_keep += len(ids); kv += 1
    tot += n_all; kept += n_keep
    trips_full.append(-(-n_all // 4)); trips_pr.append(-(-n_keep // 4))
tf = np.array(trips_full); tp = np.array(trips_pr)
print("candidates/query", tot / len(tf), "kept", kept / len(tf), "voxels hit", pv / len(tf), "kept", kv / len(tf))
m = len(tf) // 16 * 16
print("mean per-lane trips", tf.mean(), tp.mean(), " max over 16 queries (one wave):", tf[:m].reshape(-1, 16).max(1).mean(), tp[:m].reshape(-1, 16).max(1).mean())
