"""per-outer-iteration trace of the GICP stage, GPU vs oracle (FLS_LOOP_DEBUG / FLO_LOOP_DEBUG)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FLS_LOOP_DEBUG"] = "1"; os.environ["FLO_LOOP_DEBUG"] = "1"
import numpy as np
from funny_lidar_slam_amd import registration as reg, synth
from oracle import oracle as O
from tests import loopdata
src, tgt, Tt = loopdata.make_pair(job=2, n_az=300, n_t=3, n_s=2, rot_deg=(-1.0, 0.6, -4.0), trans=(-1.0, 0.7, -0.15))
fo, To, so = O.loop_match(src, tgt, np.eye(4))
T = np.eye(4); fg, sg = reg.LoopClosureMatch(src, tgt, T)
print("oracle", fo, synth.pose_error(To, Tt), "gpu", fg, synth.pose_error(T, Tt), "diff", synth.pose_error(T, To))
