"""A/B of the Gauss-Newton fan-in: ticket form (default until measured) vs flag-in-data rows (FLS_FANIN_LL=1, kernels_p2plane.hpp).
The summation order is the same in both forms, so every arm of a kind must return the SAME BITS: the pose digest, the iteration
count and n_valid are compared against the first arm, besides the timing.  For the iVox kind a 64-job fls_match_batch (8 stream
lanes: gathering workgroups of concurrent launches spin side by side) is compared as well.
usage: python tools/gpu_ab_fanin.py [ivox] [icp] [ndt] [loam] [reps=N]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth

A, B, C_ = "FLS_FANIN_LL=0 FLS_LATE_STORES=0", "FLS_FANIN_LL=0 FLS_LATE_STORES=1", "FLS_FANIN_LL=1 FLS_LATE_STORES=1"
KINDS = {
    "ivox": dict(cid=1, mode="PointToPlane_IVOX", y=reg.YAML_NCLT_IVOX, loc=False, arms=[A, B, C_, A, B, C_]),
    "icp": dict(cid=0, mode="IcpOptimized", y=reg.YAML_NCLT_ICP, loc=True, arms=[A, B, C_, A, B]),
    "ndt": dict(cid=2, mode="IncrementalNDT", y=reg.YAML_NCLT_NDT, loc=False, arms=["FLS_FANIN_LL=0", "~FLS_FUSED_TAIL=1 FLS_FANIN_LL=0", "FLS_FUSED_TAIL=1 FLS_FANIN_LL=1", "FLS_FANIN_LL=0"]),
    "loam": dict(cid=3, mode="LoamFull_KdTree", y=reg.YAML_NCLT_LOAM_FULL, loc=False, arms=[A, B, C_, A, B]),
    # reduced sizes: the 256-thread fit kernel of the iVox kind (n <= 65,536), short grids
    "ivox_small": dict(cid=1, scale=0.05, mode="PointToPlane_IVOX", y=reg.YAML_NCLT_IVOX, loc=False, arms=[A, B, C_]),
    "loam_small": dict(cid=3, scale=0.1, mode="LoamFull_KdTree", y=reg.YAML_NCLT_LOAM_FULL, loc=False, arms=[A, B, C_]),
}
# `check`: one arm on the library as built, compared with the signature the ticket form + early stores gave on the tree that passed the full GPU
# suite (profiles/r05_ll_ab_late_stores_x_fanin.log) -- the bit-identity check of a build that has no switch left to A/B against
RECORDED = {"ivox": "464154a6da", "icp": "d6ec2de0e7", "loam": "515cab1abf", "ivox_small": "2ec23863e6", "loam_small": "ead6da3ebf"}
CHECK = "check" in sys.argv[1:]
if CHECK:
    for K in KINDS.values():
        K["arms"] = [""]
args = [a for a in sys.argv[1:] if "=" not in a and a != "check"] or [k for k in KINDS if k != "ndt"]
dig = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
reps = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("reps=")), 150))
bad = 0
for kind in args:
    K = KINDS[kind]
    cfg = synth.make_config(K["cid"], scale=K.get("scale", 1.0))
    first = None
    for arm in K["arms"]:
        for k in ("FLS_FANIN_LL", "FLS_FUSED_TAIL", "FLS_LATE_STORES"):
            os.environ.pop(k, None)
        info_only = arm.startswith("~")  # an arm whose summation order differs by design (NDT's ticket-form fused tail: 16 row groups, not the solve launch's 32)
        arm = arm.lstrip("~")
        os.environ.update(dict(kv.split("=", 1) for kv in arm.split()))
        kw = dict(is_localization_mode=True) if K["loc"] else {}
        m = reg.make_matcher(K["mode"], K["y"], **kw)
        if kind.startswith("loam"):
            m.AddCloudToLocalMap([cfg["map"], cfg["corner_map"]])
            cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"], corner_cloud_=cfg["corner_scan"])
        elif kind.startswith("ivox"):
            m.AddCloudToLocalMap([cfg["map"]])
            cl = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
        else:
            m.AddCloudToLocalMap([cfg["map"]])
            cl = reg.PointcloudCluster(ordered_cloud_=cfg["scan"])
        m.UploadScan(cl)
        run, Tv = m.resident_call(np.eye(4))
        digs = []
        for _ in range(6):  # (the repeated Match is a period-2 cycle for the iVox kind, Q15: two digests)
            run(); digs.append(dig(Tv))
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
        def corr_dig():  # ids / counts / valid flags of the last Match, every slot (what the late stores write)
            parts = [np.asarray(x).astype(np.int64).ravel() for sl in ((0, 1) if kind.startswith("loam") else (0,)) for x in m.correspondences(sl)]
            return dig(np.concatenate(parts))
        sig = (tuple(digs), int(m.stats.iterations), int(m.stats.n_valid), corr_dig())
        extra = ""
        if kind.startswith("ivox"):
            rng = np.random.default_rng(5)
            Ts = []
            for j in range(64):
                T = np.eye(4); T[:3, 3] = rng.normal(scale=0.05, size=3); Ts.append(T)
            t0 = time.perf_counter()
            oks, Tb, _ = m.MatchBatch([cl] * 64, Ts, lanes=8)
            bd = dig(Tb)
            extra = f" batch64x8 {1e3 * (time.perf_counter() - t0):.1f} ms digest {bd} ok {sum(oks)}"
            sig = sig + (bd,)
        # edge cases through fls_match: a handful of points (one workgroup: the gatherer has nothing to wait for), a scan that sees nothing of the map
        far = (cfg["scan"][:200] + np.float32(5000.0)).astype(np.float32)
        for sc in (cfg["scan"][:40].copy(), far):
            if kind.startswith("loam"):
                c2 = reg.PointcloudCluster(planar_cloud_=sc, corner_cloud_=cfg["corner_scan"][:20].copy())
            elif kind.startswith("ivox"):
                c2 = reg.PointcloudCluster(planar_cloud_=sc)
            else:
                c2 = reg.PointcloudCluster(ordered_cloud_=sc)
            T = np.eye(4)
            try:
                ok = m.Match(c2, T, update_map=False)
            except Exception as e:  # (an error status is a result too: it must be the same in every arm)
                ok, T = "raised " + type(e).__name__, np.zeros((4, 4))
            sig = sig + (str(ok), dig(T), int(m.stats.iterations))
        sig = sig + (corr_dig(),)  # ... and after the edge cases
        if first is None:
            first = sig
        same = "SAME BITS" if sig == first else ("bits differ (expected: another summation order)" if info_only else "BITS DIFFER")
        bad += (sig != first) and not info_only
        if CHECK and kind in RECORDED:
            ok_rec = hashlib.sha1(repr(sig).encode()).hexdigest()[:10] == RECORDED[kind]
            same = "MATCHES THE RECORDED SIGNATURE" if ok_rec else "DIFFERS FROM THE RECORDED SIGNATURE " + RECORDED[kind]
            bad += not ok_rec
        print(f"[{kind:10s} {arm.replace('FLS_', ''):30s}] match median {1e6 * np.median(ts):7.1f} us  p10 {1e6 * np.percentile(ts, 10):7.1f}  min {1e6 * min(ts):7.1f}; iters {sig[1]} n_valid {sig[2]} "
              f"T {digs[-2]}/{digs[-1]} sig {hashlib.sha1(repr(sig).encode()).hexdigest()[:10]} {same}{extra}", flush=True)
        m.close()
print("RESULT", "ok" if bad == 0 else f"{bad} arm(s) differ")
sys.exit(1 if bad else 0)
