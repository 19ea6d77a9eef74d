#!/usr/bin/env python
"""bench.py -- scans/sec of the scan-to-map registration hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line.
For N > 1 the driver launches it through ``torch.distributed.run`` (one rank per GPU, RCCL).

A "step" is one full ``Match`` (all Gauss-Newton iterations until the reference's own stop
rule, map update excluded) of one synthetic Velodyne-64 scan (64 x 1800 = 115,200 points)
against the 1e6-point iVox map with ``LoamPointToPlaneIVOX`` semantics = BASELINE.json
configs[1].  The scan and the map are resident in HBM before the timed region.  With N GPUs
every rank registers the SAME scan against a replica of the map (identical work per GPU, so
that value(N) / N is comparable with value(1); independent jobs, no data-path collective;
"scaling": "weak"); rank 0 builds the map once, its image is broadcast (RCCL) and imported by
the other ranks; the only other collectives are the start/stop barriers, the MAX
of the per-rank times and the gather of the poses.

Extra objects on the line:
  roofline      the correspondence kernel of the headline path: algorithmic bytes per launch (8d formula, counters
                counted on the device and checked against the oracle's) / average launch duration measured with
                hipEvents attached to the kernel's own dispatch packet during the timed region; `traffic` = HBM
                bytes per launch from the committed rocprofv3 PMC passes; `detail` = what actually limits the kernel.
  pose_err_vs_oracle   |dt| [m], |dR| [rad] of the GPU result against the CPU oracle's on the same inputs.
  configs       BASELINE configs[0], [2], [3] (Optimized-ICP, Incremental-NDT, LOAM line + plane) at full size:
                scans/s, iterations, 8d roofline of their correspondence launches, CPU oracle next to them.
  mapping_mode  what the plug-in actually issues (Match + the map update inside it): ms per scan with the
                device-side AddPoints, and with the exact host path.
  inclusive_h2d fls_match from host buffers (de-interleave + PCIe copy inside the call).
  c5_batch      BASELINE configs[4]: 512 independent scan-to-map jobs (64 per GPU at 8 GPUs; distinct scans, one per job)
                block-partitioned over the ranks, fls_match_batch on 8 stream lanes, host-to-device scan upload and the
                gather of the result table inside the timed region.
  cpu_baseline  the CPU oracle (a port of the reference algorithm, pinned against the reference's own compiled code:
                tests/test_ref_pin.py) timed on this box's host cores on the headline workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

# configs[4] batch: 8 stream lanes (+ the handle's own stream) map onto hardware queues only if the runtime may open that many
# (ROCm's default is 4 per process, shared with torch's own streams); must be set before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PRE_WARM_MATCHES = 256  # untimed Matches before the warm-up steps (GPU clock ramp; see main())


def algorithmic_bytes(point_iters, probes, hits, cand, per_probe=16, per_hit=8, per_cand=12, out=24):
    """12 (src xyz) + per_probe/probe + per_hit/hit voxel + per_cand/candidate + out (indices / flags) per point-iteration."""
    return 12 * point_iters + per_probe * probes + per_hit * hits + per_cand * cand + out * point_iters


# ---------------------------------------------------------------------------------------------------------------------
# configs[4] scans: one distinct seeded scan per job, ray-cast on the host cores in parallel BEFORE HIP initialises
# ---------------------------------------------------------------------------------------------------------------------
def _cast_job(job: int):
    from funny_lidar_slam_amd import synth
    scene = synth.make_scene()
    T_gt = synth.random_pose(synth.rng_for(4, job))
    return synth.cast_scan(scene, T_gt, rng=synth.rng_for(4, job, salt=7), **synth.VELODYNE_64)


def make_batch_scans(job_ids):
    if not len(job_ids):
        return []
    workers = max(1, min(len(job_ids), (os.cpu_count() or 8) // 2, 64))
    if workers == 1:
        return [_cast_job(j) for j in job_ids]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_cast_job, list(job_ids), chunksize=max(1, len(job_ids) // (4 * workers)))


def timed_oracle(make, run, budget_s, max_reps=40):
    """median seconds of run(o) over a bounded number of repetitions on a fresh oracle from make()."""
    o = make()
    times = []
    t_start = time.perf_counter()
    for rep in range(max_reps):
        t0 = time.perf_counter()
        run(o)
        times.append(time.perf_counter() - t0)
        if rep >= 2 and time.perf_counter() - t_start > budget_s:
            break
    return float(np.median(times[1:] if len(times) > 1 else times)), len(times), o


def cpu_baseline(cfg, y, budget_s=20.0):
    """Time the CPU oracle on the headline scan / map (bounded), instrumentation off, best thread count."""
    from oracle import oracle as O

    ncpu = os.cpu_count() or 1
    best = None
    # the reference's own threading is TBB over all cores; the sequential reduction and per-query heap
    # allocations cap the scaling, so a few thread counts are tried and the best is reported
    for thr in sorted({min(ncpu, t) for t in (16, 32, 64, ncpu)}):
        O.set_threads(thr)

        def make():
            o = O.OracleMatcher(O.P2PLANE_IVOX, O.Params(max_iterations=y["optimization_iter_num"], point_to_planar_thres=y["point_to_planar_thres"],
                                                          position_converge_thres=y["position_converge_thres"],
                                                          rotation_converge_thres=y["rotation_converge_thres"]))
            o.AddCloudToLocalMap(cfg["map"])
            o.set_instrumentation(False)  # no traffic / tie bookkeeping inside the timed kNN stage
            return o

        # (nearest_points_ persists across Match calls in the reference, so one instance is re-used: same work per call)
        t, reps, o = timed_oracle(make, lambda o: o.Match(cfg["scan"], cfg["T_init"], update_map=False), budget_s / 4)
        if best is None or t < best[0]:
            best = (t, thr, o.stats.iterations, reps)
        o.close()
    O.set_threads(0)
    t, thr, iters, reps = best
    return {"value": 1.0 / t, "unit": "scans/s", "cores": thr, "threads": thr, "host_cores": ncpu, "kind": "port",
            "sample": f"{reps} Match calls (median; {budget_s:.0f} s budget over the thread counts tried) of the full 115,200-pt scan into the 1e6-pt iVox map ({iters} GN iterations each), "
                      f"OpenMP per-point stage + sequential reduction, traffic instrumentation off, best of thread counts up to {ncpu} (best: {thr})"}


def cpu_baseline_reference(cfg, y, budget_s=10.0):
    """The reference's OWN LoamPointToPlaneIVOX<double>::Match timed on the headline inputs -- SURVEY.md 8d "report both".
    oracle/_ref/libref_par.so = the reference's sources compiled verbatim against the include-shadow shim of oracle/ref_shim, with its
    parallel-STL loops (`std::for_each(std::execution::par_unseq, ...)`, loam_point_to_plane_ivox.h:262) on OpenMP threads: the shim's
    include/pstl_omp.hpp stands in for the TBB backend the reference links (CMakeLists.txt:96; TBB's headers are absent here, and
    libstdc++ alone would run those loops on one thread).  What it is and is not: the reference's loops, containers and allocations
    exactly as written, bit-identical to the serial build (tests/test_ref_pin.py); Eigen / PCL calls go through the shim's stand-ins.
    `value` = the best of a few thread counts; `one_thread` = the same library held to one thread (round 1-4's figure).
    Localization-mode instance: the same Match loop, without the map update the mapping-mode Match appends (:205-207)."""
    os.environ.setdefault("FLS_REF_PAR", "1")
    from oracle import oracle as O, ref as R
    par = R.PARALLEL and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_par.so"))
    if not par and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")) and not R.available():
        return None
    p = O.Params(max_iterations=y["optimization_iter_num"], point_to_planar_thres=y["point_to_planar_thres"], position_converge_thres=y["position_converge_thres"],
                 rotation_converge_thres=y["rotation_converge_thres"], is_localization_mode=1)
    m = R.RefMatcher(O.P2PLANE_IVOX, p)
    m.AddCloudToLocalMap(cfg["map"])
    ncpu = os.cpu_count() or 1
    counts = sorted({min(ncpu, t) for t in (1, 16, 32, 64, ncpu)}) if par else [1]
    per = {}
    iters = 0
    for thr in counts:
        if par:
            R.lib().ref_set_threads(thr)
        ts, t_start = [], time.perf_counter()
        for rep in range(12):
            t0 = time.perf_counter()
            ok, T = m.Match(cfg["scan"], cfg["T_init"])
            ts.append(time.perf_counter() - t0)
            iters = int(m.stats.iterations)
            if rep >= 2 and time.perf_counter() - t_start > budget_s / len(counts):
                break
        per[thr] = (float(np.median(ts[1:] if len(ts) > 1 else ts)), len(ts))
    m.close()
    best = min(per, key=lambda k: per[k][0])
    out = {"value": 1.0 / per[best][0], "unit": "scans/s", "cores": best, "threads": best, "host_cores": ncpu, "kind": "reference",
           "one_thread": 1.0 / per[1][0] if 1 in per else None,
           "scans_per_s_by_threads": {str(k): 1.0 / v[0] for k, v in per.items()},
           "scaling_note": "beyond 16-32 threads this build gets SLOWER (rounds 5-6: 64 threads 10.6, 256 threads 2.2-2.5 scans/s): the shim's Eigen stand-ins allocate per call and the "
                           "allocator serialises them -- a property of the shim build, not of the reference with real Eigen + TBB; the best thread count is what `value` reports",
           "sample": f"{sum(v[1] for v in per.values())} Match calls (median per thread count, {budget_s:.0f} s budget) of the full 115,200-pt scan into the 1e6-pt iVox map "
                     f"({iters} GN iterations), the reference's own LoamPointToPlaneIVOX<double> compiled verbatim (oracle/ref_shim), shim Eigen / PCL; "
                     + ("its parallel-STL loops on OpenMP threads (pstl_omp.hpp in place of the TBB backend the reference links)" if par
                        else "serial PSTL backend (libref_par.so absent) -- 1 core")}
    return out


def grid_counters(map_xyz, query_xyz, cell):
    """8d counters of a 27-cell uniform grid (cell = sqrt of the squared-distance gate) for kd-tree kinds: probes, hit cells, candidates."""
    inv = 1.0 / cell
    mk = np.floor(map_xyz.astype(np.float64) * inv).astype(np.int64)
    lo = mk.min(0) - 2
    span = (mk.max(0) - lo + 3).astype(np.int64)
    key = lambda k: ((k[:, 2] - lo[2]) * span[1] + (k[:, 1] - lo[1])) * span[0] + (k[:, 0] - lo[0])
    uk, cnt = np.unique(key(mk), return_counts=True)
    qk = np.floor(query_xyz.astype(np.float64) * inv).astype(np.int64)
    ok = np.all((qk >= lo + 1) & (qk < lo + span - 1), axis=1)
    qk = qk[ok]
    hits = cand = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                k = key(qk + np.array([dx, dy, dz]))
                pos = np.searchsorted(uk, k)
                pos[pos >= uk.size] = uk.size - 1
                found = uk[pos] == k
                hits += int(found.sum())
                cand += int(cnt[pos][found].sum())
    return 27 * int(query_xyz.shape[0]), hits, cand


def bench_other_configs(reg, synth, util, O, budget_s=6.0):
    """BASELINE configs[0], [2], [3] at full size: GPU resident Match, 8d roofline of the correspondence launches, CPU oracle."""
    out = {}
    cases = [("configs[0] Optimized-ICP (16x900 scan, 50k-pt fixed map, localization)", 0, "IcpOptimized", reg.YAML_NCLT_ICP, True),
             ("configs[2] Incremental-NDT (64x1800 scan, 1.0 m voxels, 1e6-pt map)", 2, "IncrementalNDT", reg.YAML_NCLT_NDT, False),
             ("configs[3] LOAM frontend (57,600 surf + 7,680 corner points, point-to-line + point-to-plane)", 3, "LoamFull_KdTree", reg.YAML_NCLT_LOAM_FULL, False)]
    for name, cid, mode, y, loc in cases:
        cfg = synth.make_config(cid)
        maps = [cfg["map"]] + ([cfg["corner_map"]] if "corner_map" in cfg else [])
        corner = cfg.get("corner_scan")
        m = reg.make_matcher(mode, y, is_localization_mode=loc)
        m.AddCloudToLocalMap(maps)
        cl = util.cluster_for(mode, cfg["scan"], corner)
        # What the reference's Match does, with its input already in device memory: IcpOptimized / IncrementalNDT VoxelGrid their source cloud
        # FIRST (icp_optimized.h:57, incremental_ndt.h:231-232), so `scans_per_s` times fls_scan_upload_raw + fls_match_resident -- the raw
        # cloud resident, the pcl::VoxelGrid (bit-identical device filter) inside every timed call.  The round-1..4
        # figure -- the filter done once at upload, only the iterations timed -- stays as `resident_filtered_*`.
        filters_inside = mode in ("IcpOptimized", "IncrementalNDT")
        ts_raw = None
        if filters_inside:
            m.UploadScanRaw(cl)
            run_raw, Tv_raw = m.resident_call(np.eye(4))
            for _ in range(5):
                run_raw()
            ts_raw = []
            for _ in range(30):
                t = time.perf_counter(); run_raw(); ts_raw.append(time.perf_counter() - t)
            T_raw = np.array(Tv_raw)
            iters_raw = int(m.stats.iterations)
        m.UploadScan(cl)
        run, Tv = m.resident_call(np.eye(4))
        for _ in range(5):
            run()
        ts = []
        for _ in range(30):
            t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
        if filters_inside:
            assert np.array_equal(T_raw, np.array(Tv)) and iters_raw == int(m.stats.iterations), "filter inside the call must not change the result"
        m.set_profiling(True)
        for _ in range(10):
            run()
        ms, nl, _ = m.kernel_time()
        m.set_profiling(False)
        iters = int(m.stats.iterations)
        T_gpu = np.array(Tv)
        # fls_match from HOST buffers (upload + source VoxelGrid + Match): the device filter in std::sort's order (default since round 4,
        # bit-identical to the host filter), the exact host filter, and the round-2/3 device filter (leaf sums in point-index order, A/B)
        from_host = None
        if mode in ("IcpOptimized", "IncrementalNDT"):
            from_host = {}
            prev = os.environ.get("FLS_DEVICE_VOXELGRID")
            for label, env in (("device_filter_exact_default", "1"), ("host_filter_exact", "0"), ("device_filter_index_order_ab", "2")):
                os.environ["FLS_DEVICE_VOXELGRID"] = env
                m2 = reg.make_matcher(mode, y, is_localization_mode=loc)
                m2.AddCloudToLocalMap(maps)
                t2 = []
                for k in range(25):
                    Tm = np.eye(4)
                    t = time.perf_counter(); m2.Match(cl, Tm, update_map=False); t2.append(time.perf_counter() - t)
                from_host[label] = 1e6 * float(np.median(t2[5:]))
                m2.close()
            if prev is None:
                os.environ.pop("FLS_DEVICE_VOXELGRID", None)
            else:
                os.environ["FLS_DEVICE_VOXELGRID"] = prev
        # CPU oracle: result (pose error, counters) + bounded timing
        O.set_threads(min(os.cpu_count() or 1, 64))

        def make():
            o = util.oracle_for(mode, y, loc)
            o.AddCloudToLocalMap(*maps)
            return o

        t_cpu, reps, o = timed_oracle(make, lambda o: o.Match(cfg["scan"], np.eye(4), src1=corner, update_map=False), budget_s, max_reps=12)
        ok_ref, T_ref = o.Match(cfg["scan"], np.eye(4), src1=corner, update_map=False)
        cnt = o.counters()
        dt, dr = synth.pose_error(T_gpu, T_ref)
        n_pts = int(m.stats.n_source) + int(m.stats.n_source_corner)
        # 8d algorithmic bytes of ONE iteration at the final pose (kd-tree kinds: the 27-cell grid of the survey's formula)
        if mode == "IncrementalNDT":
            pi = int(cnt.point_iters) // max(int(o.stats.iterations), 1)
            bytes_iter = algorithmic_bytes(pi, int(cnt.probes) // max(int(o.stats.iterations), 1), int(cnt.hit_voxels) // max(int(o.stats.iterations), 1), 0,
                                           per_probe=16, per_hit=24 + 48 + 4, per_cand=0, out=4)
        else:
            gate = float(np.sqrt(y["point_search_thres"]))
            q = (cfg["scan"].astype(np.float64) @ T_ref[:3, :3].T + T_ref[:3, 3]).astype(np.float32)
            if mode == "IcpOptimized":  # Match VoxelGrids the scan (icp_optimized.h:57) and the map (:187) first
                src = np.asarray(O.voxel_grid(cfg["scan"], y["source_cloud_filter_size"]))[:, :3]
                q = (src.astype(np.float64) @ T_ref[:3, :3].T + T_ref[:3, 3]).astype(np.float32)
                pr, hi, ca = grid_counters(np.asarray(O.voxel_grid(cfg["map"], y["local_map_cloud_filter_size"]))[:, :3], q, gate)
                bytes_iter = algorithmic_bytes(q.shape[0], pr, hi, ca, out=8)
            else:
                pr, hi, ca = grid_counters(cfg["map"], q, gate)
                qc = (corner.astype(np.float64) @ T_ref[:3, :3].T + T_ref[:3, 3]).astype(np.float32)
                pr2, hi2, ca2 = grid_counters(cfg["corner_map"], qc, gate)
                bytes_iter = algorithmic_bytes(q.shape[0] + qc.shape[0], pr + pr2, hi + hi2, ca + ca2)
        avg_launch_s = (ms / 1e3) / max(nl, 1)
        ach = bytes_iter / avg_launch_s / 1e9 if nl else 0.0
        # HBM bytes per launch and what limits the kernel, from the committed PMC passes of this kind's correspondence kernel
        # (tools/prof_round5.sh kinds -> tools/make_kernel_traffic_json.py -> profiles/traffic_<kernel>.json; rocprofv3 cannot collect counters from inside this process)
        traffic, tdetail = None, None
        try:
            tname = {"IncrementalNDT": "ndt_lanes", "IcpOptimized": "icp_knn_fit", "LoamFull_KdTree": "grid_knn_dual"}[mode]
            with open(os.path.join(ROOT, "profiles", "traffic_%s.json" % tname)) as f:
                tj = json.load(f)
            traffic = tj.get("hbm_bytes_per_launch")
            tdetail = {k: tj[k] for k in ("kernel", "trace_avg_launch_us", "measured_hbm_GBs", "l2_hit_rate", "valu_busy_pct", "wave_wait_pct", "waves_per_launch") if k in tj}
            tdetail["source"] = "profiles/traffic_%s.json" % tname
        except (OSError, KeyError, ValueError):
            pass
        t_whole = float(np.median(ts_raw)) if filters_inside else float(np.median(ts))
        out[f"configs[{cid}]"] = {
            "workload": name, "scans_per_s": 1.0 / t_whole, "match_us": 1e6 * t_whole,
            "timed_call": ("fls_scan_upload_raw once, then fls_match_resident: the source pcl::VoxelGrid + every iteration per call (the whole reference Match, input resident)"
                           if filters_inside else "fls_scan_upload once, then fls_match_resident (this kind's Match has no source filter)"),
            "resident_filtered_match_us": 1e6 * float(np.median(ts)), "resident_filtered_scans_per_s": 1.0 / float(np.median(ts)),
            "gn_iterations": iters,
            "converged": bool(m.stats.converged), "source_points": n_pts, "pose_err_vs_oracle_m_rad": [dt, dr],
            "match_from_host_buffers_us": from_host,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "detail": tdetail,
                         "algorithmic_bytes_per_iteration": bytes_iter, "correspondence_launch_us": 1e6 * avg_launch_s,
                         "note": "8d formula; kd-tree kinds: counters of the survey's 27-cell grid (cell = sqrt(gate)) at the final pose, the bracket covers one "
                                 "iteration's correspondence launch(es)"
                                 + (" -- IncrementalNDT: since round 6 that launch also holds the fan-in and the Gauss-Newton tail (one launch per iteration; it was a "
                                    "12.5 us correspondence launch + a 7.7 us solve launch), so `frac` fell from 0.10 to ~0.07 while the iteration got shorter" if mode == "IncrementalNDT" else "")},
            "cpu_baseline": {"value": 1.0 / t_cpu, "unit": "scans/s", "cores": min(os.cpu_count() or 1, 64), "threads": min(os.cpu_count() or 1, 64), "host_cores": os.cpu_count() or 1, "kind": "port",
                             "sample": f"{reps} oracle Match calls (median), {int(o.stats.iterations)} iterations each"},
        }
        o.close()
        m.close()
    O.set_threads(0)
    return out


def bench_mapping_mode(reg, synth, cfg, n_scans=6):
    """Match + the map update the reference performs inside Match (what the adapter issues), device path and host path."""
    scene = cfg["scene"]
    rng = synth.rng_for(1, 123)
    Tgt = cfg["T_gt"].copy()
    scans = []
    for k in range(n_scans):
        scans.append(synth.cast_scan(scene, Tgt, rng=rng, **synth.VELODYNE_64))
        Tgt = Tgt @ synth.random_pose(rng, 0.5, 0.5)
    res = {}
    for label, env in (("device_addpoints", "1"), ("host_addpoints", "0")):
        os.environ["FLS_IVOX_DEVICE_UPDATE"] = env
        m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        m.AddCloudToLocalMap([cfg["map"]])
        guess = np.eye(4)
        t_match, t_both = [], []
        for k, scan in enumerate(scans):
            cl = reg.PointcloudCluster(planar_cloud_=scan)
            m.UploadScan(cl)
            T = guess.copy(); m.MatchResident(T, update_map=False)  # warm (buffers), Match only
            T = guess.copy(); t = time.perf_counter(); m.MatchResident(T, update_map=False); t_match.append(time.perf_counter() - t)
            T = guess.copy(); t = time.perf_counter(); m.MatchResident(T, update_map=True); t_both.append(time.perf_counter() - t)
            guess = T
        res[label] = {"ms_per_scan_match_plus_update": 1e3 * float(np.median(t_both[1:])), "ms_map_update_only": 1e3 * float(np.median(np.array(t_both[1:]) - np.array(t_match[1:]))),
                      "ms_match_only": 1e3 * float(np.median(t_match[1:])), "device_batches": m.map_size(103), "refused_batches": m.map_size(104),
                      "map_points_after": m.map_size()}
        m.close()
    os.environ.pop("FLS_IVOX_DEVICE_UPDATE", None)
    # the production pipeline hands Match a 0.5 m voxel-filtered planar cloud (preprocessing.cpp:236-237), not the raw scan:
    # the insert rule then keeps the map sparse.  Same six poses, scans filtered with the product's device VoxelGrid
    # (fls_debug_voxel_grid), fls_match from HOST buffers with update_map = 1 -- the adapter's real call.
    try:
        import ctypes as C
        from funny_lidar_slam_amd import _lib

        declined = [0]

        def filt(c, leaf=0.5):
            # pcl::VoxelGrid of the preprocessing thread (preprocessing.cpp:236-237) on the device, bit-identical to the reference's (round 4);
            # what the device declines (introsort's heap-sort case) takes the exact host filter -- counted
            a = np.ascontiguousarray(c, np.float32)
            out = np.zeros((a.shape[0], 4), np.float32)
            n_out = C.c_size_t(0)
            fp = C.POINTER(C.c_float)
            rc = _lib.lib().fls_voxel_grid_cloud(0, 1, a.ctypes.data_as(fp), a.shape[0], a.shape[1], np.float32(leaf), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
            if rc != 0:
                declined[0] += 1
                return np.ascontiguousarray(reg.VoxelGridCloud(c, leaf, on_device=False)[:, :3])
            return np.ascontiguousarray(out[: n_out.value, :3])

        fscans = [filt(sc) for sc in scans]
        m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        m.AddCloudToLocalMap([filt(cfg["map"])])
        guess = np.eye(4)
        tb, its = [], []
        for sc in fscans:
            T = guess.copy(); t = time.perf_counter(); m.Match(reg.PointcloudCluster(planar_cloud_=sc), T, update_map=True); tb.append(time.perf_counter() - t)
            its.append(int(m.stats.iterations))
            guess = T
        res["filtered_planar_cloud_0p5m"] = {"scan_points": int(np.median([s.shape[0] for s in fscans])), "map_points_after": m.map_size(),
                                             "ms_per_scan_match_plus_update_from_host_buffers": 1e3 * float(np.median(tb[1:])),
                                             "gn_iterations": its, "device_batches": m.map_size(103), "refused_batches": m.map_size(104),
                                             "preprocessing_filters_declined_by_the_device": declined[0], "preprocessing_filters": n_scans + 1}
        m.close()
    except Exception as e:
        res["filtered_planar_cloud_0p5m"] = {"error": repr(e)[:200]}
    try:
        res["incremental_ndt"] = bench_ndt_mapping_mode(reg, synth)
    except Exception as e:  # never at the expense of the rest of the line
        res["incremental_ndt"] = {"error": repr(e)[:200]}
    try:
        res.update(bench_kd_mapping_mode(reg, synth))
    except Exception as e:
        res["icp_optimized"] = {"error": repr(e)[:200]}
    res["note"] = (f"{n_scans} consecutive scans (0.5 m / 0.5 deg steps), scan resident; medians over scans 1..; the scans that follow a map update run more "
                   "iterations against a map the insert rule has densified near the sensor, so ms_match_only here is not the headline step")
    return res


def bench_kd_mapping_mode(reg, synth, n_scans=6):
    """IcpOptimized / LoamFull in mapping mode as the ROS adapter issues them: fls_match from HOST buffers with update_map = 1; every
    frame passes the keyframe gate, so every Match ends in AddCloudToLocalMap = deque push + VoxelGrid of the concatenated deque +
    kd-tree (here: cell grid) rebuild (icp_optimized.h:165-189, loam_full_kdtree.h:65-104).  Three settings: the round-2 host path
    (host filter, host grid build + upload), the default (exact host filter, grid built on the device) and the opt-in device path
    (deque resident on the device, map-side VoxelGrid + grid build on the device).  The deque is pre-filled so that the timed
    keyframes see a local map of production size."""
    out = {}
    keys = ("FLS_DEVICE_GRID_BUILD", "FLS_DEVICE_VOXELGRID")
    prev = {k: os.environ.get(k) for k in keys}
    cases = [("icp_optimized", "IcpOptimized", dict(reg.YAML_NCLT_ICP), 0, 30), ("loam_full", "LoamFull_KdTree", dict(reg.YAML_NCLT_LOAM_FULL), 3, 24)]
    for label, mode, y, cid, prefill in cases:
        cfg = synth.make_config(cid)
        scene = cfg["scene"]
        rng = synth.rng_for(cid, 777)
        lid = synth.VELODYNE_16 if cid == 0 else synth.VELODYNE_64
        Tgt = np.eye(4)
        frames = []
        for k in range(n_scans + 1):
            scan = synth.cast_scan(scene, Tgt, rng=rng, **lid)
            corner = None
            if cid == 3:
                corner = synth.cast_edge_scan(scene, Tgt, 7680, rng)
                scan = scan[::2].copy()
            frames.append((scan, corner, Tgt.copy()))
            step = np.eye(4)
            step[:3, :3] = synth.so3_exp(np.deg2rad([0.0, 0.0, 3.0]))
            step[:3, 3] = [1.2, 0.1, 0.0]  # beyond the 1.0 m keyframe gate
            Tgt = Tgt @ step
        world = lambda c, T: (c.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        res = {}
        for setting, env in (("host_path_round2", ("0", "0")), ("host_filter_device_grid_build_round3_default", ("1", "0")), ("default_device_filter_exact", ("1", "1")),
                             ("device_filter_index_order_ab", ("1", "2"))):
            os.environ.update(dict(zip(keys, env)))
            m = reg.make_matcher(mode, y)
            s0, c0, T0 = frames[0]
            init = [world(s0, T0)] + ([world(c0, T0)] if c0 is not None else [])
            for _ in range(prefill + 1):  # the frontend's first AddCloudToLocalMap, then the pre-fill (same clouds: the filter merges them)
                m.AddCloudToLocalMap(init)
            t_match, t_both, upd = [], [], 0
            for scan, corner, Tk in frames[1:]:
                cl = util_cluster(reg, mode, scan, corner)
                g = Tk.copy()  # prediction = ground truth of the frame (an IMU-grade guess)
                T = g.copy(); t = time.perf_counter(); m.Match(cl, T, update_map=False); t_match.append(time.perf_counter() - t)
                T = g.copy(); t = time.perf_counter(); m.Match(cl, T, update_map=True); t_both.append(time.perf_counter() - t)
                upd += int(m.stats.map_updated)
            res[setting] = {"ms_match_only_from_host_buffers": 1e3 * float(np.median(t_match[1:])), "ms_match_plus_keyframe_update": 1e3 * float(np.median(t_both[1:])),
                            "ms_keyframe_update_only": 1e3 * float(np.median(np.array(t_both[1:]) - np.array(t_match[1:]))), "keyframes": upd,
                            "map_points_after": [m.map_size(0)] + ([m.map_size(1)] if cid == 3 else []),
                            "grids_built_on_device": m.map_size(114), "map_filters_on_device": m.map_size(115), "map_filters_on_host": m.map_size(116)}
            m.close()
        a, b = res["host_path_round2"]["ms_keyframe_update_only"], res["default_device_filter_exact"]["ms_keyframe_update_only"]
        res["keyframe_update_speedup_device_vs_host"] = a / b if b > 0 else None
        res["deque_frames_at_timing"] = prefill + 1
        out[label] = res
    for k, v in prev.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return out


def util_cluster(reg, mode, scan, corner):
    if mode == "IcpOptimized":
        return reg.PointcloudCluster(ordered_cloud_=scan)
    return reg.PointcloudCluster(planar_cloud_=scan, corner_cloud_=corner)


def bench_loop_closure(reg, O):
    """LoopClosure::Match (src/slam/loop_closure.cpp:233-267: 4-resolution NDT + GICP + fitness) on two synthetic sub-maps:
    fls_loop_match against the CPU oracle's restatement (OpenMP in its kNN stages only)."""
    from tests import loopdata
    src, tgt, Tt = loopdata.make_pair(job=1, n_az=450, n_t=5, n_s=3)
    ts = []
    for k in range(4):
        T = np.eye(4)
        t = time.perf_counter(); f, st = reg.LoopClosureMatch(src, tgt, T); ts.append(time.perf_counter() - t)
    t = time.perf_counter(); fo, To, so = O.loop_match(src, tgt, np.eye(4)); t_cpu = time.perf_counter() - t
    from funny_lidar_slam_amd import synth
    dt, dr = synth.pose_error(T, To)
    et, er = synth.pose_error(T, Tt)
    return {"workload": f"source sub-map {src.shape[0]} pts, target sub-map {tgt.shape[0]} pts (keyframe clouds VoxelGrid 0.2 m, merged), guess = identity, true offset 0.95 m / 3 deg",
            "ms_per_match": 1e3 * float(np.median(ts[1:])), "cpu_oracle_ms": 1e3 * t_cpu, "fitness": f, "pose_err_vs_oracle_m_rad": [dt, dr], "pose_err_vs_truth_m_rad": [et, er],
            "ndt_iterations": list(st.ndt_iterations), "ndt_evaluations": list(st.ndt_evaluations), "ndt_source_points": list(st.ndt_source_points),
            "gicp_outer_inner_evaluations": [st.gicp_iterations, st.gicp_inner_iterations, st.gicp_evaluations], "gicp_correspondences": st.gicp_correspondences,
            "note": "per-point sums on the device (ndt_p2d_kernel, gicp_cov / corr / fdf kernels), six-parameter optimisers + leaf statistics + exact VoxelGridCloud filters on the host"}


def _dig(d, path):
    for k in path.split("/"):
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return float(d) if isinstance(d, (int, float)) and not isinstance(d, bool) else None


# short name -> (path in the bench line, "lower"/"higher" is better).  The LAST key of the printed line (`legs`) is this table evaluated: every leg a
# round can regress on, flat and short, so that the 2,000-character tail the driver keeps of the line still carries all of them, and
# tools/bench_diff.py can compare a new line with the previous round's record leg by leg (VERDICT r5 weak #12 / next #1b).
LEGS = {
    "headline_us": ("ms_per_step", "lower", 1e3), "knn_us": ("roofline/avg_launch_us", "lower", 1.0), "knn_frac": ("roofline/frac", "higher", 1.0),
    "fit_us": ("roofline_fit_kernel/avg_launch_us", "lower", 1.0),
    "icp_sps": ("configs/configs[0]/scans_per_s", "higher", 1.0), "icp_iters_us": ("configs/configs[0]/resident_filtered_match_us", "lower", 1.0),
    "icp_corr_us": ("configs/configs[0]/roofline/correspondence_launch_us", "lower", 1.0), "icp_frac": ("configs/configs[0]/roofline/frac", "higher", 1.0),
    "icp_host_us": ("configs/configs[0]/match_from_host_buffers_us/device_filter_exact_default", "lower", 1.0),
    "icp_host_idx_us": ("configs/configs[0]/match_from_host_buffers_us/device_filter_index_order_ab", "lower", 1.0),
    "icp_cpu_sps": ("configs/configs[0]/cpu_baseline/value", "higher", 1.0),
    "ndt_sps": ("configs/configs[2]/scans_per_s", "higher", 1.0), "ndt_iters_us": ("configs/configs[2]/resident_filtered_match_us", "lower", 1.0),
    "ndt_corr_us": ("configs/configs[2]/roofline/correspondence_launch_us", "lower", 1.0), "ndt_frac": ("configs/configs[2]/roofline/frac", "higher", 1.0),
    "ndt_host_us": ("configs/configs[2]/match_from_host_buffers_us/device_filter_exact_default", "lower", 1.0),
    "ndt_host_idx_us": ("configs/configs[2]/match_from_host_buffers_us/device_filter_index_order_ab", "lower", 1.0),
    "ndt_cpu_sps": ("configs/configs[2]/cpu_baseline/value", "higher", 1.0),
    "loam_sps": ("configs/configs[3]/scans_per_s", "higher", 1.0), "loam_corr_us": ("configs/configs[3]/roofline/correspondence_launch_us", "lower", 1.0),
    "loam_frac": ("configs/configs[3]/roofline/frac", "higher", 1.0), "loam_cpu_sps": ("configs/configs[3]/cpu_baseline/value", "higher", 1.0),
    "map_ivox_ms": ("mapping_mode/device_addpoints/ms_per_scan_match_plus_update", "lower", 1.0),
    "map_ivox_upd_ms": ("mapping_mode/device_addpoints/ms_map_update_only", "lower", 1.0),
    "map_ivox_prod_ms": ("mapping_mode/filtered_planar_cloud_0p5m/ms_per_scan_match_plus_update_from_host_buffers", "lower", 1.0),
    "map_ndt_ms": ("mapping_mode/incremental_ndt/default_device_update_device_filters_exact/ms_per_scan_match_plus_update", "lower", 1.0),
    "map_ndt_idx_ms": ("mapping_mode/incremental_ndt/device_update_device_filters_index_order_ab/ms_per_scan_match_plus_update", "lower", 1.0),
    "map_icp_kf_ms": ("mapping_mode/icp_optimized/default_device_filter_exact/ms_keyframe_update_only", "lower", 1.0),
    "map_icp_kf_idx_ms": ("mapping_mode/icp_optimized/device_filter_index_order_ab/ms_keyframe_update_only", "lower", 1.0),
    "map_loam_kf_ms": ("mapping_mode/loam_full/default_device_filter_exact/ms_keyframe_update_only", "lower", 1.0),
    "map_loam_kf_idx_ms": ("mapping_mode/loam_full/device_filter_index_order_ab/ms_keyframe_update_only", "lower", 1.0),
    "loop_ms": ("loop_closure/ms_per_match", "lower", 1.0), "h2d_us": ("inclusive_h2d/match_us", "lower", 1.0),
    "c5_sps": ("c5_batch/scans_per_s", "higher", 1.0), "c5n_sps": ("c5_batch_native/scans_per_s", "higher", 1.0),
    "cpu_sps": ("cpu_baseline/value", "higher", 1.0), "cpu_ref_sps": ("cpu_baseline_ref/value", "higher", 1.0),
}


def legs_from_line(line: dict) -> dict:
    """The flat leg table of a bench line (also of the lines of earlier rounds, which did not print one)."""
    out = {}
    for name, (path, _, scale) in LEGS.items():
        v = _dig(line, path)
        if v is not None:
            out[name] = float(f"{v * scale:.5g}")
    return out


def baseline_metric():
    """BASELINE.json's metric string, verbatim (it travels with the repository); the ASCII spelling if the file is missing"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "scans/sec (64-ring x 1800 pts -> 1e6-pt iVox map), SE(3) err vs CPU ref"


def bench_ndt_mapping_mode(reg, synth, n_scans=6):
    """configs[2] as the ROS adapter issues it: fls_match from HOST buffers with update_map = 1 (source VoxelGrid + Match + the
    reference's in-Match AddCloud), three settings: everything the round-1 way on the host, the device map update behind the
    exact host filters (the round-3 default), the device filters in std::sort order (default since round 4: bit-identical, the whole chain on the
    device) and the round-2/3 device filters (leaf sums in point-index order) for A/B."""
    cfg = synth.make_config(2)
    rng = synth.rng_for(2, 321)
    Tgt = cfg["T_gt"].copy()
    scans = []
    for k in range(n_scans):
        scans.append(synth.cast_scan(cfg["scene"], Tgt, rng=rng, **synth.VELODYNE_64))
        Tgt = Tgt @ synth.random_pose(rng, 0.3, 0.2)
    out = {}
    keys = ("FLS_NDT_DEVICE_UPDATE", "FLS_DEVICE_VOXELGRID")
    prev = {k: os.environ.get(k) for k in keys}
    for label, env in (("host_update_host_filters", ("0", "0")), ("device_update_host_filters_round3_default", ("1", "0")), ("default_device_update_device_filters_exact", ("1", "1")),
                       ("device_update_device_filters_index_order_ab", ("1", "2"))):
        os.environ.update(dict(zip(keys, env)))
        m = reg.make_matcher("IncrementalNDT", reg.YAML_NCLT_NDT)
        m.AddCloudToLocalMap([cfg["map"]])
        guess = np.eye(4)
        tm, tb = [], []
        for scan in scans:
            cl = reg.PointcloudCluster(ordered_cloud_=scan)
            T = guess.copy(); t = time.perf_counter(); m.Match(cl, T, update_map=False); tm.append(time.perf_counter() - t)
            T = guess.copy(); t = time.perf_counter(); m.Match(cl, T, update_map=True); tb.append(time.perf_counter() - t)
            guess = T
        out[label] = {"ms_match_only": 1e3 * float(np.median(tm[1:])), "ms_per_scan_match_plus_update": 1e3 * float(np.median(tb[1:])),
                      "device_batches": m.map_size(109), "refused_batches": m.map_size(110), "voxels_after": m.map_size()}
        m.close()
    for k, v in prev.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return out


def bench_c5_native(m, reg, scans, n_gpus, share, table_torch, lanes=8):
    """BASELINE configs[4] through fls_replicas_match_batch: one process, device list 0..N-1 (N x device 0 with FLS_BENCH_SHARE_DEVICE=1)."""
    from funny_lidar_slam_amd import batch

    devices = [0] * n_gpus if share else list(range(n_gpus))
    t0 = time.perf_counter()
    rs = m.Replicas(devices)
    t_create = time.perf_counter() - t0
    clusters = [reg.PointcloudCluster(planar_cloud_=s) for s in scans]
    T0s = [np.eye(4)] * len(clusters)
    rs.MatchBatch(clusters[:16 * n_gpus], T0s[:16 * n_gpus], lanes=lanes)  # lane creation, buffer growth
    rs.MatchBatch(clusters, T0s, lanes=lanes)                              # one untimed pass (host pages of every scan touched)
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        oks, Tb, sb = rs.MatchBatch(clusters, T0s, lanes=lanes)
        reps.append(time.perf_counter() - t0)
    tb = float(np.median(reps))
    rows = np.stack([batch.pack_result(Tb[k], oks[k], sb[k].iterations, sb[k].n_valid, sb[k].sum_res) for k in range(len(clusters))])
    out = {"jobs": len(clusters), "devices": devices, "lanes_per_gpu": lanes, "scans_per_s": len(clusters) / tb, "ms_total": 1e3 * tb,
           "ms_passes": [1e3 * t for t in reps], "converged_jobs": int(np.sum(rows[:, 16] == 1.0)), "replica_set_create_ms": 1e3 * t_create,
           "import_ms_per_device": [float(t) for t in rs.import_ms()],
           "note": "fls_replicas_match_batch: one process, one host thread + one replica handle per device, block partition, no collective; "
                   "uploads from host memory inside the timed region"}
    if table_torch is not None and table_torch.shape == rows.shape:
        out["table_equals_torch_form_bitwise"] = bool(np.array_equal(rows, table_torch))
    rs.close()
    # what replicating the map costs, measured where one GPU is enough: a second handle on device 0 filled by the
    # device-to-device image copy (default) and by the round-3 export blob + import (FLS_REPLICAS_VIA_BLOB=1); both must answer alike
    try:
        sub = clusters[:32]
        rep = {}
        for label, env in (("device_image_copy", "0"), ("export_blob_plus_import_round3", "1")):
            os.environ["FLS_REPLICAS_VIA_BLOB"] = env
            t0 = time.perf_counter()
            r2 = m.Replicas([0, 0])
            t_all = time.perf_counter() - t0
            oks2, T2, s2 = r2.MatchBatch(sub, T0s[:len(sub)], lanes=lanes)
            same = all(np.array_equal(T2[k], Tb[k]) and s2[k].n_valid == sb[k].n_valid for k in range(len(sub)))
            rep[label] = {"set_create_ms": 1e3 * t_all, "replica_fill_ms": float(r2.import_ms()[1]), "first_32_jobs_equal_the_owner_table": bool(same)}
            r2.close()
        out["map_replication_one_more_handle_on_this_gpu"] = rep
    except Exception as e:  # noqa: BLE001 -- an extra, never the reason a bench line is lost
        out["map_replication_one_more_handle_on_this_gpu"] = {"error": repr(e)}
    finally:
        os.environ.pop("FLS_REPLICAS_VIA_BLOB", None)
    return out


def self_launch(n: int) -> int:
    """Re-run this command line under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous, a free port)."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks through torch.distributed.run (port {port})", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the configs[4] batch measurement")
    ap.add_argument("--no-native-batch", action="store_true", help="skip c5_batch_native (configs[4] through fls_replicas_match_batch, one process / N devices)")
    ap.add_argument("--no-extras", action="store_true", help="skip configs[0,2,3], mapping_mode, inclusive_h2d (N = 1 extras)")
    ap.add_argument("--batch-jobs", type=int, default=512, help="configs[4] jobs in total at 8 GPUs (64 per GPU); scaled by n_gpus / 8 below 8 GPUs unless --batch-jobs-total")
    ap.add_argument("--batch-jobs-total", type=int, default=0, help="configs[4]: total number of jobs (default: 512 at N = 1 and at N = 8, 64 * N otherwise)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl == RCCL; gloo + FLS_BENCH_SHARE_DEVICE=1 exercises the N > 1 code path on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Called bare (`python bench.py --gpus 8`, the shape of the driver's N = 1 command): launch the ranks ourselves -- one process
        # per GPU through torch.distributed.run, RCCL over xGMI -- and hand its exit code back.  Never a quiet 1-GPU run.
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (torch.distributed.run sets RANK / WORLD_SIZE) the process group is ALWAYS created, also at world size 1: that is how
    # the RCCL branch -- nccl init with device_id, CUDA-tensor broadcast / all_gather / all_reduce, the nccl + gloo group mix -- runs on a
    # 1-GPU box (tests/test_gpu_batch_ranks.py::test_rccl_path_at_world_size_1; VERDICT r4 missing #1)
    distributed = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    n_gpus = world if distributed else 1

    from funny_lidar_slam_amd import batch, synth

    # configs[4] inputs first (host cores, forked workers: nothing GPU-side may be initialised yet)
    n_jobs = args.batch_jobs_total or (512 if n_gpus in (1, 8) else 64 * n_gpus)
    my_scans, all_scans, job_range = [], [], (0, 0)
    if not args.no_batch:
        job_range = batch.partition(n_jobs, n_gpus, rank)
        t_gen = time.perf_counter()
        # every rank casts ITS block only; rank 0, which also drives the native one-process form over ALL jobs (c5_batch_native), collects
        # the other blocks over a host-side gloo group below (round 6: rank 0 cast all 512 scans itself before, 22 s on its own)
        my_scans = make_batch_scans(range(*job_range))
        t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist

    share = os.environ.get("FLS_BENCH_SHARE_DEVICE", "0") == "1"  # test hook: every rank on device 0 (1-GPU box)
    dev = 0 if (share or not distributed) else local_rank
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"  # where the collective payloads live
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend="gloo")
    host_group = None  # CPU-side group: the scan gather below, and the idle barrier of the native-batch leg
    if distributed and not args.no_batch and not args.no_native_batch:
        host_group = dist.new_group(backend="gloo")
        t_g = time.perf_counter()
        all_scans = batch.gather_scans(my_scans, n_jobs, dst=0, group=host_group)
        t_gather = time.perf_counter() - t_g
    elif not args.no_batch and not args.no_native_batch:
        all_scans, t_gather = my_scans, 0.0
    if args.gpus != n_gpus:
        # a launcher that started a different number of ranks than --gpus says: refuse rather than report a line for the wrong N
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to run", file=sys.stderr)
        if distributed:
            dist.destroy_process_group()
        sys.exit(2)

    from funny_lidar_slam_amd import _lib, registration as reg

    if _lib.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X (gfx950); the HIP path has no CPU fallback")
    torch.cuda.set_device(dev)

    y = reg.YAML_NCLT_IVOX
    cfg = synth.make_config(1, job=0, with_map=(rank == 0))  # identical scan on every rank; the map is built once, on rank 0
    m = reg.make_matcher("PointToPlane_IVOX", y, device_id=dev)
    t_map = time.perf_counter()
    if rank == 0:
        m.AddCloudToLocalMap([cfg["map"]])
    map_bcast = None
    if distributed:
        # rank 0 builds the map, its image travels in one broadcast, every other GPU imports it
        # (round 5: the DEVICE image travels -- exported straight into the collective's buffer, imported from it; rounds 2-4 shipped the host blob:
        # ExportMap 90 ms + ImportMap 78 ms per rank.  FLS_BENCH_BLOB_BROADCAST=1 keeps that form for A/B.)
        if os.environ.get("FLS_BENCH_BLOB_BROADCAST", "0") == "1":
            te0 = time.perf_counter()
            blob = m.ExportMap() if rank == 0 else None
            export_ms = 1e3 * (time.perf_counter() - te0)
            tb0 = time.perf_counter()
            blob = batch.broadcast_blob(blob, src=0, device=coll_dev)
            tb1 = time.perf_counter()
            if rank != 0:
                m.ImportMap(blob)
            tr = {"image_MB": blob.size / 1e6, "export_ms": export_ms, "broadcast_ms": 1e3 * (tb1 - tb0), "import_ms": 1e3 * (time.perf_counter() - tb1), "buffer": "host blob"}
            del blob
        else:
            tr = batch.broadcast_map_image(m, src=0, device=coll_dev)
        ti = torch.tensor([tr["import_ms"], tr["export_ms"], tr["broadcast_ms"]], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ti, op=dist.ReduceOp.MAX)  # rank 0 imports nothing, the others export nothing: the slowest rank of each
        map_bcast = {"image_MB": tr["image_MB"], "buffer": tr["buffer"], "export_ms_rank0": float(ti[1].item()), "broadcast_ms": float(ti[2].item()),
                     "import_ms_max_over_ranks": float(ti[0].item()),
                     "note": "the flat device image (points | brick directory | cells) written into the collective's own buffer by fls_map_image_export and taken "
                             "from it by fls_map_image_import; importing ranks are read-only replicas"}
    t_map = time.perf_counter() - t_map
    cluster = reg.PointcloudCluster(planar_cloud_=cfg["scan"])
    m.UploadScan(cluster)  # inputs resident in HBM before the timed region

    run, T_work = m.resident_call(cfg["T_init"], update_map=False)  # every step registers from the same initial guess

    def step():
        rc = run()
        if rc < 0:
            raise RuntimeError(f"fls_match_resident failed: {rc}")
        return rc == 0, T_work

    ok_first, T_first = step()  # the first Match of the handle: what the oracle's first Match is compared with
    T_first = np.array(T_first)
    # Clock ramp (set-up, NOT warm-up steps): the GPU idles for seconds while the host builds the 1e6-point map, and 5 warm-up steps
    # are 0.6 ms -- the timed region would run on a power state that is still ramping (measured: 121 us per step at W = 5 / K = 20
    # against 110 us at W = 100 / K = 200, kNN launch 18.6 vs 17.6 us).  A SLAM front-end registers scans continuously, so the
    # steady state is the number that means something: PRE_WARM_MATCHES untimed Matches (~30 ms) precede the W warm-up steps.
    # They count as calls of the handle in the call-k bookkeeping below.
    for _ in range(PRE_WARM_MATCHES - 2):
        step()
    m.set_profiling(True)   # two of the untimed Matches run bracketed: the library creates its event ring on the first profiled Match,
    step(); step()          # which must not be a timed one (two, not one: the parity of the call index is kept, Q15)
    m.set_profiling(False)
    m.kernel_time()
    for _ in range(args.warmup):
        step()
    # start / stop hipEvents are attached to every correspondence-kernel launch of every EVENT_EVERY-th step of the
    # timed region (hipExtLaunchKernelGGL: the kernel's own execution time); they are settled after the region.  A bracketed step
    # costs ~36 us more than a plain one (measured, tools/loop_overhead.py: 125 us / step without events, 134 with every 4th, 128 with
    # every 16th), so the sampling is kept sparse: steps 16, 48, ... = 1 step / 3 launches at K = 20, 2 steps / 6 launches at K = 50
    # (launch durations repeat to +-1 us).
    EVENT_EVERY = 32
    m.kernel_time()  # reset the accumulators
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # (mid-period steps, not step 0: the first Match after the barrier + synchronize starts on an idle GPU and its launches read 2-3 us
    # longer than the steady state the other 19 steps run in -- 17.4-18.6 us against 15.6-16.0 us over 60 launches, tools/gpu_ab.py)
    bracketed = {k for k in range(args.steps) if k % EVENT_EVERY == EVENT_EVERY // 2} or {args.steps // 2}
    for k in range(args.steps):
        if k in bracketed:
            m.set_profiling(True)
            ok, T = step()
            m.set_profiling(False)
        else:
            ok, T = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms_kernel, launches, point_iters = m.kernel_time()
    iters = m.stats.iterations
    n_valid_head = m.stats.n_valid
    T_head = np.array(T)
    # algorithmic-traffic counters: one extra (untimed) Match with the counting kernel variant
    m.set_profiling(False, counters=True)
    step()
    probes, hits, cand = m.traffic_counters()
    m.set_profiling(False, counters=False)
    assert m.stats.iterations == iters

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        row = batch.pack_result(T_head, ok, m.stats.iterations, m.stats.n_valid, m.stats.sum_res)
        table = batch.gather_results(row[None, :], world, batch.RESULT_WIDTH, device=coll_dev)  # one job per rank per step
        assert table.shape == (world, batch.RESULT_WIDTH)
        assert bool(np.all(table == table[0])), "identical jobs on identical GPUs (imported map image) must give bit-identical results"

    c5 = None
    if not args.no_batch:
        # configs[4]: job ids block-partitioned (batch.partition), every job its own scan, 8 stream lanes per GPU
        # (tools/gpu_batch.py, 256 jobs: 1 lane 3.8k, 2: 7.0k, 4: 7.7k, 8: 10.2k, 16: 10.8k scans/s)
        lanes = 8
        clusters = [reg.PointcloudCluster(planar_cloud_=s) for s in my_scans]
        T0s = [np.eye(4)] * len(clusters)
        warm = clusters[: min(len(clusters), 16)]
        for w in range(2):  # warm-up: lane creation, buffer growth, clocks back up after the host-side preparation; then one untimed pass
            if warm:        # over the whole batch (every scan's host pages touched once: the timed passes read 700 MB of scans)
                m.MatchBatch(warm if w == 0 else clusters, T0s[: len(warm)] if w == 0 else T0s, lanes=lanes)
        reps = []
        for _ in range(3):  # median of three passes over the whole batch
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            oks, Tb, sb = m.MatchBatch(clusters, T0s, lanes=lanes) if clusters else ([], np.zeros((0, 4, 4)), [])
            rows = (np.stack([batch.pack_result(Tb[k], oks[k], sb[k].iterations, sb[k].n_valid, sb[k].sum_res) for k in range(len(clusters))])
                    if clusters else np.zeros((0, batch.RESULT_WIDTH)))
            tab = batch.gather_results(rows, n_jobs, batch.RESULT_WIDTH, device=coll_dev if distributed else None)
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            tb = time.perf_counter() - tb
            if distributed:
                tm = torch.tensor([tb], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                tb = float(tm.item())
            reps.append(tb)
        tb = float(np.median(reps))
        assert tab.shape == (n_jobs, batch.RESULT_WIDTH)
        c5 = {"jobs": n_jobs, "jobs_per_gpu": n_jobs // n_gpus, "lanes_per_gpu": lanes, "scans_per_s": n_jobs / tb, "ms_total": 1e3 * tb,
              "ms_passes": [1e3 * t for t in reps], "converged_jobs": int(np.sum(tab[:, 16] == 1.0)),
              "gn_iterations_hist": {str(int(v)): int(c) for v, c in zip(*np.unique(tab[:, 17], return_counts=True))},
              "distinct_poses": int(len({tuple(np.round(r[:16], 9)) for r in tab})), "scan_generation_s": t_gen,
              "scan_gather_to_rank0_s": (t_gather if not args.no_native_batch else None),
              "note": "one distinct seeded scan per job (seed 20241022 + 4000 + job), all against ONE map; fls_match_batch: per-job scan upload from host "
                      "memory and the result gather are inside the timed region"}
        if map_bcast is not None:
            c5["map_image_broadcast"] = map_bcast

    # configs[4] in its NATIVE form (SURVEY.md 8e "one process + N host threads"; include/fls_reg.h fls_replicas_*): rank 0's process holds one
    # replica handle per device 0..N-1 (the owner's map image imported per device on that device's own host thread), the same n_jobs jobs are
    # block-partitioned like batch.partition and run through fls_match_batch per device; results land in the caller's arrays, no collective.
    # The other ranks idle on a CPU-side (gloo) barrier meanwhile -- an RCCL barrier would spin a kernel on the GPUs being measured.
    c5n = None
    if not args.no_batch and not args.no_native_batch:
        idle = host_group
        if rank == 0:
            try:
                c5n = bench_c5_native(m, reg, all_scans, n_gpus, share, tab if c5 is not None else None)
            except Exception as e:  # never at the expense of the line
                c5n = {"error": repr(e)[:300]}
        if distributed:
            dist.barrier(group=idle)
    del all_scans

    if rank == 0:
        from oracle import oracle as O
        from tests import util
        total_scans = args.steps * n_gpus
        value = total_scans / elapsed
        per_launch_bytes = algorithmic_bytes(point_iters=cfg["scan"].shape[0] * iters, probes=probes, hits=hits, cand=cand) / max(iters, 1)
        avg_launch_s = (ms_kernel / 1e3) / max(launches, 1)
        achieved = per_launch_bytes / avg_launch_s / 1e9 if launches else 0.0
        dt, dr = synth.pose_error(T_head, cfg["T_gt"])
        # HBM bytes per launch / what limits the kernel, from the committed PMC passes (rocprofv3 cannot collect counters from
        # inside this process; gpurun keeps --pmc runs separate from everything else): profiles/traffic_ivox_knn.json
        traffic, detail = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_ivox_knn.json")) as f:
                tj = json.load(f)
            traffic = float(tj["hbm_bytes_per_launch"])
            detail = {k: tj[k] for k in ("l2_read_bytes_per_launch", "valu_busy_pct", "wave_wait_pct", "valu_wave_instructions_per_launch",
                                         "instruction_floor_us", "source") if k in tj}
            if traffic and avg_launch_s:
                detail["measured_hbm_GBs"] = traffic / avg_launch_s / 1e9
        except (OSError, KeyError, ValueError):
            traffic = None
        line = {
            "metric": baseline_metric(),
            "value": value, "unit": "scans/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "distributed_backend": (args.backend if distributed else None),
            "protocol": {"version": 3, "pre_warm_matches": PRE_WARM_MATCHES, "event_bracketed_steps": "mid-period, one step in 32",
                         "note": "version 3 since round 3 (256 untimed pre-warm Matches, mid-period event brackets, call-k oracle comparison); rounds 1-2 "
                                 "lines (protocol 1 / 2: no pre-warm, bracket on step 0 / every 8th step) are not directly comparable"},
            "config": {"workload": "BASELINE configs[1]: Velodyne-64 synthetic scan (64x1800 = 115,200 pts), point-to-plane "
                                   "(LoamPointToPlaneIVOX semantics, YAML config_nclt.yaml) into a 1e6-pt iVox map, 1 scan per GPU per step "
                                   "(the same scan on every GPU)",
                       "scan_points": int(cfg["scan"].shape[0]), "map_points": int(m.map_size()),
                       "gn_iterations": int(iters), "converged": bool(ok), "pose_err_vs_gt_m_rad": [dt, dr],
                       "pre_warm_matches": PRE_WARM_MATCHES},
            "roofline": {"bound": "hbm", "kernel": "ivox_knn_kernel<4>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/traffic_ivox_knn.json (separate rocprofv3 --pmc passes, bytes per launch)",
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_us": 1e6 * avg_launch_s,
                         "launches_timed": int(launches), "device_counters": {"probes": int(probes), "hit_voxels": int(hits), "cand_points": int(cand)}},
        }
        if detail:
            line["roofline"]["detail"] = detail
        # the OTHER kernel of an iteration: per-point plane fit + Jacobian + the 29 normal-equation sums + the Gauss-Newton tail in the last
        # workgroup.  Not a streaming kernel: ~600 FP64 flop and 112 B of gathered rows per point, then a serial fan-in + one-wave tail.
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_p2plane_fit_solve.json")) as f:
                fj = json.load(f)
            n_pts = int(cfg["scan"].shape[0])
            fit_bytes = n_pts * (12 + 32 + 5 * 16 + 7 * 8 + 1)  # xyz + id row + five gathered map points + J s | |d| row (Q1 state) + flag, per point
            fit_flop = n_pts * 600
            lus = float(fj["trace_avg_launch_us"])
            line["roofline_fit_kernel"] = {
                "kernel": "p2plane_fit_solve_kernel", "avg_launch_us": lus, "launch_geometry": "225 workgroups x 512 threads = 1.76 waves / SIMD, 127 VGPRs",
                "algorithmic_bytes_per_launch": fit_bytes, "achieved_GBs_by_algorithmic_bytes": fit_bytes / (lus * 1e-6) / 1e9, "frac_of_hbm_peak": fit_bytes / (lus * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "fp64_flop_per_launch": fit_flop, "achieved_fp64_TFLOPs": fit_flop / (lus * 1e-6) / 1e12,
                "traffic": fj.get("hbm_bytes_per_launch"), "measured_hbm_GBs": fj.get("measured_hbm_GBs"), "valu_busy_pct": fj.get("valu_busy_pct"),
                "wave_wait_pct": fj.get("wave_wait_pct"), "l2_hit_rate": fj.get("l2_hit_rate"),
                "bound": "latency: ~8 us of parallel per-point work, then the fan-in of 225 partial rows and a single-wave Gauss-Newton tail (serial)",
                "source": "profiles/traffic_p2plane_fit_solve.json (tools/prof_round5.sh kinds: kernel trace + four PMC passes)"}
        except (OSError, KeyError, ValueError):
            pass
        if c5 is not None:
            line["c5_batch"] = c5
        if c5n is not None:
            line["c5_batch_native"] = c5n
        if n_gpus == 1:
            # the CPU oracle on the same inputs: pose error of the headline result, and the device counters against the oracle's
            # (the timed steps re-register the same scan on ONE handle, and nearest_points_ persists from Match to Match in the
            # reference -- Q15: a point without candidates keeps its previous list -- so the steady-state step equals the oracle's
            # SECOND and later calls on one instance, not its first; the first-call parity is what the gpu tests check)
            o = util.oracle_for("PointToPlane_IVOX", y)
            o.AddCloudToLocalMap(cfg["map"])
            ok_ref, T_ref = o.Match(cfg["scan"], cfg["T_init"], update_map=False)
            edt, edr = synth.pose_error(T_first, T_ref)
            first_iters = int(o.stats.iterations)
            # call k of the handle vs call k of the oracle: the handle has run 1 + warmup + steps Matches when T_head is read and one more
            # (the counting variant) when the device counters are read; the oracle is driven through the same number of calls.
            # Re-registering one scan on one handle is NOT a fixed point: nearest_points_ persists (Q15) and the state enters a
            # period-2 cycle after ~4 calls (poses 1.7e-5 m apart, candidate counts 13,415,502 / 13,415,539 on this workload) -- which is
            # what round 2's "27th call vs the oracle's 3rd" comparison tripped over.  Beyond 64 calls an EVEN number of oracle calls is
            # skipped (same phase of the cycle); tests/test_gpu_parity.py::test_repeated_match_..._full_size asserts call-by-call equality.
            n_head = 1 + PRE_WARM_MATCHES + args.warmup + args.steps
            n_oracle = n_head + 1
            while n_oracle > 64:
                n_oracle -= 2
            k_head = n_oracle - 1  # the oracle call the timed step's result is compared with
            T_refk, ok_refk, it_refk, nv_refk = T_ref, ok_ref, first_iters, int(o.stats.n_valid)
            for call in range(2, n_oracle + 1):
                ok_k, T_k = o.Match(cfg["scan"], cfg["T_init"], update_map=False)
                if call == k_head:
                    T_refk, ok_refk, it_refk, nv_refk = T_k, ok_k, int(o.stats.iterations), int(o.stats.n_valid)
            oc = o.counters()
            sdt, sdr = synth.pose_error(T_head, T_refk)
            counters_equal = (int(oc.probes), int(oc.hit_voxels), int(oc.cand_points)) == (int(probes), int(hits), int(cand))
            line["pose_err_vs_oracle"] = {"dt_m": edt, "dR_rad": edr, "same_iterations": bool(first_iters == iters), "same_return": bool(ok_ref == ok_first),
                                          "what": "first Match of the handle vs the oracle's first Match (identical inputs and state)",
                                          "timed_step_vs_oracle_same_call_index": {
                                              "dt_m": sdt, "dR_rad": sdr, "handle_calls": n_head, "oracle_calls": n_oracle,
                                              "same_iterations": bool(it_refk == iters), "same_return": bool(ok_refk == ok), "same_n_valid": bool(nv_refk == int(n_valid_head)),
                                              "device_counters_equal_oracle_counters": bool(counters_equal),
                                              "note": "the timed region re-registers one resident scan on one handle; nearest_points_ persists (Q15) and the state runs "
                                                      "into a period-2 cycle, so call k is compared with the oracle's call k (an even number of calls skipped beyond 64)"},
                                          "oracle_counters_same_call": {"probes": int(oc.probes), "hit_voxels": int(oc.hit_voxels), "cand_points": int(oc.cand_points)}}
            o.close()
            if not args.no_extras:
                t = time.perf_counter()
                try:
                    line["configs"] = bench_other_configs(reg, synth, util, O)
                    line["mapping_mode"] = bench_mapping_mode(reg, synth, cfg)
                    try:
                        line["loop_closure"] = bench_loop_closure(reg, O)
                    except Exception as e:
                        line["loop_closure"] = {"error": repr(e)[:200]}
                    # the boundary handing over host buffers: de-interleave into pinned staging + one H2D copy inside the call
                    ts = []
                    for _ in range(5):
                        Th = np.eye(4); m.Match(cluster, Th, update_map=False)
                    for _ in range(50):
                        Th = np.eye(4); t1 = time.perf_counter(); m.Match(cluster, Th, update_map=False); ts.append(time.perf_counter() - t1)
                    line["inclusive_h2d"] = {"match_us": 1e6 * float(np.median(ts)), "scans_per_s": 1.0 / float(np.median(ts)),
                                             "note": "fls_match with the scan as a host buffer (12 B / point): never `value`"}
                except Exception as e:  # the extras must never cost the headline line
                    line["extras_error"] = repr(e)[:300]
                line["extras_wall_s"] = time.perf_counter() - t
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cfg, y)
                try:  # the reference's own compiled code beside the port (never at the expense of the line)
                    ref_line = cpu_baseline_reference(cfg, y)
                    if ref_line is not None:
                        line["cpu_baseline_ref"] = ref_line
                except Exception as e:
                    line["cpu_baseline_ref"] = {"error": repr(e)[:200]}
        line["legs"] = legs_from_line(line)  # LAST key on purpose: the driver keeps the tail of the line (see LEGS)
        print(json.dumps(line))
    m.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
