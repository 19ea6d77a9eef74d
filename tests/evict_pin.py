"""IVoxMap::AddPoints at a tiny capacity, one batch with thousands of in-loop evictions (src/ivox_map/ivox_map.cpp:122-143: "evict inside the insert
loop"), many of them of voxels that a later point of the SAME batch re-creates -- the corner the device-side eviction walk resolves since round 4
(csrc/kernels_ivox_update.hpp ivox_evict_select).  Three witnesses of the same batch:
  * the compiled reference (oracle/_ref/libref.so: the reference's own ivox_map.cpp), in a fresh process (its first-call flag is function-static),
  * the CPU oracle (oracle/flo_common.h IVoxMap::AddPoints),
  * a ten-line sequential model in Python (also tells how many voxels were evicted / re-created, so that the test is known to hit the corner).
Used by tests/test_ref_pin.py; `python -m tests.evict_pin <out.npz>` runs the reference side, `python -m tests.evict_pin --golden` rewrites
tests/golden/ref_ivox_recreate.npz (needs libref.so)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAPACITY = 60
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_ivox_recreate.npz")


def make_cloud():
    """3,000 points over 140 voxels of the 0.5 m grid (capacity 60): a slowly drifting window of recently used voxels + returns to old ones."""
    rng = np.random.default_rng(20240924)
    n, universe = 3000, 140
    seq = np.empty(n, np.int64)
    centre = 0.0
    for i in range(n):
        centre += 0.03
        if rng.random() < 0.25:
            seq[i] = rng.integers(0, universe)                       # anywhere: often a voxel evicted a while ago, or about to be
        else:
            seq[i] = int(centre + rng.normal(0.0, 12.0)) % universe  # near the drifting window
    kx, ky, kz = seq % 14 - 7, (seq // 14) % 10 - 5, np.zeros(n, np.int64)
    keys = np.stack([kx, ky, kz], 1)
    pts = keys * 0.5 + rng.uniform(-0.2, 0.2, (n, 3))  # |jitter| < 0.25: round(p * 2) is the key
    pts = pts.astype(np.float32)
    assert np.array_equal(np.round(pts * np.float32(2.0)).astype(np.int64), keys)
    return pts, keys


def sequential_model(keys, capacity):
    """the reference's loop on keys alone: returns (LRU order front first as key tuples, {key: [point indices]}, evictions, re-creations)"""
    lru, vox, evicted_once, n_evict, n_recreate = [], {}, set(), 0, 0
    for i, k in enumerate(map(tuple, keys)):
        if k not in vox:
            vox[k] = [i]
            lru.insert(0, k)
            n_recreate += k in evicted_once
            if len(vox) >= capacity:
                b = lru.pop()
                del vox[b]
                evicted_once.add(b)
                n_evict += 1
        else:
            vox[k].append(i)
            lru.remove(k)
            lru.insert(0, k)
    return lru, vox, n_evict, n_recreate


def run_oracle():
    from funny_lidar_slam_amd import registration as reg
    from tests import util
    pts, _ = make_cloud()
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    o.set_ivox_capacity(CAPACITY)
    o.AddCloudToLocalMap(pts)
    out = dict(points=o.map_dump(0).astype(np.float32), n_points=o.map_size(), n_voxels=o.map_voxels())
    o.close()
    return out


def run_ref():
    from funny_lidar_slam_amd import registration as reg
    from oracle import ref as R
    from tests import util
    pts, _ = make_cloud()
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m = R.RefMatcher(o.kind, o.params)
    o.close()
    m.set_ivox_capacity(CAPACITY)
    m.AddCloudToLocalMap(pts)
    xyzi, keys = m.map_dump(0)
    out = dict(points_lru_order=xyzi[:, :3].astype(np.float32), keys_lru_order=keys.astype(np.int32), n_points=m.map_size(0), n_voxels=m.map_voxels())
    m.close()
    return out


def run_ref_subprocess():
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "out.npz")
        subprocess.check_call([sys.executable, "-m", "tests.evict_pin", path], cwd=ROOT)
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


def check(ref, ora):
    """ref (live or golden) vs the Python model (LRU order, every voxel's points in order) vs the oracle (surviving points, in insertion order)"""
    pts, keys = make_cloud()
    lru, vox, n_evict, n_recreate = sequential_model(keys, CAPACITY)
    assert n_evict > 500 and n_recreate > 100, (n_evict, n_recreate)  # the batch does hit the corner, many times
    want_idx = [i for k in lru for i in vox[k]]            # the reference's dump order: list front to back, points in insertion order
    assert int(ref["n_voxels"]) == len(lru) == CAPACITY - 1
    assert np.array_equal(ref["keys_lru_order"], np.array([k for k in lru for _ in vox[k]], np.int32))
    assert np.array_equal(ref["points_lru_order"], pts[want_idx])
    assert int(ora["n_voxels"]) == len(lru) and int(ora["n_points"]) == len(want_idx) == int(ref["n_points"])
    assert np.array_equal(ora["points"], pts[sorted(want_idx)])  # (the oracle dumps in insertion-id order)
    return n_evict, n_recreate


if __name__ == "__main__":
    if sys.argv[1] == "--golden":
        r = run_ref_subprocess()
        np.savez_compressed(GOLDEN, **r)
        print("wrote", GOLDEN, {k: v.shape for k, v in r.items()})
    else:
        np.savez(sys.argv[1], **run_ref())
