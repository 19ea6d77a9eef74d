"""Timing of the LOAM feature front-end: GPU (device time by hipEvents + wall time of the two C-ABI calls, host copies
included) vs the CPU oracle, on synthetic Velodyne-64 frames (about 115k raw returns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import features, synth
from oracle import oracle as O
from tests.test_oracle_features import VELO64

scene = synth.make_scene()
raw = synth.cast_raw_scan(scene, np.eye(4), rng=synth.rng_for(3, 0, 9), **synth.VELODYNE_64)
g = features.FeatureFrontEnd(1800, 64, VELO64["horizontal_resolution"], 4.0, 100.0, 1.0, 0.1)
o = O.OracleFeatures(**VELO64)
for _ in range(3):
    g.project(raw); g.extract()
tp, te, wp, we = [], [], [], []
for _ in range(30):
    t0 = time.perf_counter(); n = g.project(raw); t1 = time.perf_counter(); nc, npl = g.extract(); t2 = time.perf_counter()
    a, b = g.times_ms()
    tp.append(a); te.append(b); wp.append(1e3 * (t1 - t0)); we.append(1e3 * (t2 - t1))
cp, ce = [], []
for _ in range(10):
    t0 = time.perf_counter(); o.Project(raw); t1 = time.perf_counter(); o.ExtractFeatures(); t2 = time.perf_counter()
    cp.append(1e3 * (t1 - t0)); ce.append(1e3 * (t2 - t1))
md = lambda v: float(np.median(v))
print(f"raw {raw.shape[0]} pts -> ordered {n}, corners {nc}, planar {npl}")
print(f"GPU  Project: device {md(tp):.3f} ms (H2D of the raw cloud + 3 kernels), call {md(wp):.3f} ms (with D2H of the cluster fields)")
print(f"GPU  Extract: device {md(te):.3f} ms (2 kernels), call {md(we):.3f} ms (with D2H + host gather of the clouds)")
print(f"CPU  oracle : Project {md(cp):.3f} ms, ExtractFeatures {md(ce):.3f} ms (1 thread, as the reference: seq projection, per-sector sorts)")
print(f"speed-up device-time: project {md(cp)/md(tp):.1f}x, extract {md(ce)/md(te):.1f}x; call-time: {(md(cp)+md(ce))/(md(wp)+md(we)):.1f}x")
