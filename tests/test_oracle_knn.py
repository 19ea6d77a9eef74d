"""Exact-kNN pieces of the oracle: kd-tree vs brute force vs scipy.cKDTree, VoxelGrid vs a numpy re-derivation."""
import numpy as np
from scipy.spatial import cKDTree

from funny_lidar_slam_amd import synth
from oracle import oracle as O


def test_kdtree_matches_bruteforce_and_scipy():
    rng = np.random.default_rng(7)
    m = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
    q = rng.uniform(-22, 22, (400, 3)).astype(np.float32)
    idx, d2 = O.kdtree_knn(m, q, 5)
    tree = cKDTree(m.astype(np.float64))
    dd, ii = tree.query(q.astype(np.float64), k=5)
    for i in range(q.shape[0]):
        bi, bd = O.knn_bruteforce(m, q[i], 5)
        assert np.array_equal(idx[i], bi) and np.array_equal(d2[i], bd)  # exact, incl. (d2, idx) order
        assert np.all(np.diff(d2[i]) >= 0)
        # scipy works in double: same sets unless float rounding creates a near tie
        if not np.array_equal(np.sort(idx[i]), np.sort(ii[i])):
            assert np.min(np.abs(np.diff(np.sort(np.concatenate([dd[i] ** 2, d2[i].astype(np.float64)]))))) < 1e-5
        assert np.allclose(np.sqrt(d2[i]), dd[i], rtol=1e-5, atol=1e-6)


def test_kdtree_duplicates_and_small_maps():
    m = np.zeros((7, 3), np.float32)
    m[3:] = 1.0
    idx, d2 = O.kdtree_knn(m, np.zeros((1, 3), np.float32), 5)
    assert list(idx[0]) == [0, 1, 2, 3, 4]  # ties resolved to the lower index
    idx, d2 = O.kdtree_knn(m[:3], np.zeros((1, 3), np.float32), 5)
    assert list(idx[0]) == [0, 1, 2, -1, -1]


def voxel_grid_np(c, leaf):
    inv = np.float32(1.0) / np.float32(leaf)
    mn = c[:, :3].min(0); mx = c[:, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int64); max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(c[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    out = []
    for leaf_id in np.unique(idx):
        sel = c[idx == leaf_id]
        out.append(sel.astype(np.float64).mean(0))
    return np.asarray(out), np.unique(idx)


def test_voxel_grid_semantics():
    rng = np.random.default_rng(8)
    c = np.concatenate([rng.uniform(-10, 10, (4000, 3)), rng.uniform(0, 1, (4000, 1))], 1).astype(np.float32)
    out = O.voxel_grid(c, 0.4)
    ref, _ = voxel_grid_np(c, 0.4)
    assert out.shape[0] == ref.shape[0]              # one point per occupied leaf, ascending leaf index
    assert np.allclose(out, ref, atol=2e-5)          # centroid of ALL fields (xyz + intensity), float accumulation
    # identity (up to order) when every leaf holds one point
    thin = O.voxel_grid(out, 0.4)
    assert thin.shape == out.shape
    assert np.array_equal(np.sort(thin.view([('', np.float32)] * 4), 0), np.sort(out.view([('', np.float32)] * 4), 0))
    assert O.voxel_grid(np.zeros((0, 3), np.float32), 0.4).shape[0] == 0


def test_voxel_grid_overflow_returns_input():
    # PCL: "Leaf size is too small for the input dataset" -> output = input (Appendix D)
    c = np.array([[0, 0, 0], [3000, 3000, 3000], [1, 2, 3]], np.float32)
    out = O.voxel_grid(c, 0.001)
    assert np.array_equal(out[:, :3], c)


def test_ivox_knn_against_bruteforce_python():
    """iVox 19-voxel 5-NN of the oracle (through a 1-iteration Match) vs a direct numpy restatement."""
    cfg = synth.make_config(1, scale=0.02)
    o = O.OracleMatcher(O.P2PLANE_IVOX, O.Params(max_iterations=1, point_to_planar_thres=0.1, position_converge_thres=0.005,
                                                  rotation_converge_thres=0.001))
    o.AddCloudToLocalMap(cfg["map"])
    o.Match(cfg["scan"], np.eye(4), update_map=False)
    ids, cnt, valid = o.correspondences()
    m = cfg["map"]
    keys = np.round(m * np.float32(2.0)).astype(np.int64)  # Pos2Grid: round half away (np.round is half-even: exclude .5 cases)
    frac = np.abs(m * np.float32(2.0) - np.trunc(m * np.float32(2.0)))
    assert not np.any(frac == 0.5)
    table = {}
    for i, k in enumerate(map(tuple, keys)):
        table.setdefault(k, []).append(i)
    nearby = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (1, 1, 0), (-1, 1, 0), (1, -1, 0), (-1, -1, 0),
              (1, 0, 1), (-1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 1), (0, -1, 1), (0, 1, -1), (0, -1, -1)]
    s = cfg["scan"]
    checked = 0
    for i in range(0, s.shape[0], 7):
        q = s[i]  # T = identity: transformed point == source point
        fq = np.abs(q * np.float32(2.0) - np.trunc(q * np.float32(2.0)))
        if np.any(fq == 0.5):
            continue
        k = np.round(q * np.float32(2.0)).astype(np.int64)
        cand = []
        for d in nearby:
            cand += table.get((k[0] + d[0], k[1] + d[1], k[2] + d[2]), [])
        if not cand:
            assert cnt[i] == 0
            continue
        diff = m[cand] - q
        d2 = diff[:, 0] * diff[:, 0] + (diff[:, 1] * diff[:, 1] + diff[:, 2] * diff[:, 2])
        order = np.argsort(d2, kind="stable")[:5]
        if len(order) >= 2 and (d2[order[0]] == d2[order[1]] or (len(d2) > 5 and np.sort(d2)[4] == np.sort(d2)[5])):
            continue
        want = sorted(cand[j] for j in order)
        got = sorted(int(x) for x in ids[i] if x >= 0)
        assert got == want, i
        assert ids[i, 0] == cand[order[0]]  # slot 0 is the nearest
        checked += 1
    assert checked > 200
