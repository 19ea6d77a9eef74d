"""The map-side pcl::VoxelGrid of the kd-tree kinds on the clouds bench.py's mapping_mode legs filter (the concatenated keyframe deque of
IcpOptimized: 31 x 14,400 points, leaf 0.4 m; LoamFull planar: 25 x 57,600, leaf 0.4 m; LoamFull corner: 25 x 7,680, leaf 0.2 m), timed on
its own through fls_debug_voxel_grid_timed: wall-clock per call with the buffers allocated, for whatever build / switches the environment
selects (FLS_REG_LIB, FLS_ES_*).  `python tools/gpu_vg_large.py [reps] [which,...]`; FLS_ES_DEBUG=1 adds the exact sort's stage stamps."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib, synth


def deque_cloud(cid, prefill, n_new=2):
    cache = f"/tmp/fls_vg_large_{cid}_{prefill}_{n_new}.npz"  # (the A/B script runs this tool once per build / switch setting)
    if os.path.exists(cache):
        z = np.load(cache)
        return z["planar"], (z["corner"] if "corner" in z.files else None)
    planar, corner = _deque_cloud(cid, prefill, n_new)
    np.savez(cache, planar=planar, **({"corner": corner} if corner is not None else {}))
    return planar, corner


def _deque_cloud(cid, prefill, n_new):
    cfg = synth.make_config(cid)
    scene = cfg["scene"]
    rng = synth.rng_for(cid, 777)
    lid = synth.VELODYNE_16 if cid == 0 else synth.VELODYNE_64
    Tgt = np.eye(4)
    world = lambda c, T: (c.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    planar, corner = [], []
    for k in range(n_new + 1):
        scan = synth.cast_scan(scene, Tgt, rng=rng, **lid)
        if cid == 3:
            corner.append(world(synth.cast_edge_scan(scene, Tgt, 7680, rng), Tgt))
            scan = scan[::2].copy()
        planar.append(world(scan, Tgt))
        step = np.eye(4)
        step[:3, :3] = synth.so3_exp(np.deg2rad([0.0, 0.0, 3.0]))
        step[:3, 3] = [1.2, 0.1, 0.0]
        Tgt = Tgt @ step
    cat = lambda L: np.ascontiguousarray(np.concatenate([L[0]] * (prefill + 1) + L[1:], axis=0))
    return cat(planar), (cat(corner) if corner else None)


def timed(cloud, leaf, reps):
    L = _lib.lib()
    n = cloud.shape[0]
    pts = np.zeros((n, 4), np.float32)
    pts[:, :3] = cloud[:, :3]
    ms = np.zeros(reps)
    n_out = C.c_size_t(0)
    rc = L.fls_debug_voxel_grid_timed(0, pts.ctypes.data_as(C.POINTER(C.c_float)), n, 4, np.float32(leaf), reps, ms.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_out))
    return {"n": n, "leaf": leaf, "rc": rc, "n_out": int(n_out.value), "ms_first": float(ms[0]), "ms_median_rest": float(np.median(ms[1:])) if reps > 1 else None,
            "ms_min": float(ms.min())}


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["icp", "loam_planar", "loam_corner", "scan"]
    out = {"lib": os.environ.get("FLS_REG_LIB", "default"), "env": {k: v for k, v in os.environ.items() if k.startswith("FLS_ES")}}
    if "icp" in which:
        c, _ = deque_cloud(0, 30)
        out["icp_deque"] = timed(c, 0.4, reps)
    if "loam_planar" in which or "loam_corner" in which:
        p, c = deque_cloud(3, 24)
        if "loam_planar" in which:
            out["loam_planar_deque"] = timed(p, 0.4, reps)
        if "loam_corner" in which:
            out["loam_corner_deque"] = timed(c, 0.2, reps)
    if "scan" in which:  # the source filter's size class, for reference (the fused one-stream form)
        cfg = synth.make_config(2)
        out["ndt_source_scan"] = timed(np.ascontiguousarray(cfg["scan"][:, :3]), 0.2, reps)
    print(json.dumps(out))
