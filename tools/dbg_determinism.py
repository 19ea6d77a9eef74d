"""hunt for a rare nondeterminism: (a) the device VoxelGrid on one cloud N times (output hash), (b) fresh IcpOptimized handle + map + Match N times (pose hash)"""
import ctypes as C, hashlib, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from funny_lidar_slam_amd import _lib, registration as reg, synth
from tests import util
what = sys.argv[1]; N = int(sys.argv[2])
cfgs = [synth.make_config(0, job=j, scale=1.0) for j in range(4)]
if what == "vg":
    L = _lib.lib()
    for name, cloud, leaf in (("map50k", cfgs[0]["map"], 0.4), ("scan3", cfgs[3]["scan"], 0.4), ("scan0", cfgs[0]["scan"], 0.4)):
        a = np.zeros((cloud.shape[0], 4), np.float32); a[:, :3] = cloud[:, :3]
        out = np.zeros((a.shape[0], 4), np.float32); n_out = C.c_size_t(0); fp = C.POINTER(C.c_float)
        cnt = collections.Counter()
        for r in range(N):
            out[:] = 0
            rc = L.fls_debug_voxel_grid(0, a.ctypes.data_as(fp), a.shape[0], 4, np.float32(leaf), out.ctypes.data_as(fp), out.shape[0], C.byref(n_out))
            cnt[(rc, n_out.value, hashlib.sha1(out[:n_out.value].tobytes()).hexdigest()[:12])] += 1
        print("vg", name, cloud.shape[0], dict(cnt), flush=True)
else:
    mode, y = "IcpOptimized", reg.YAML_NCLT_ICP
    cl = util.cluster_for(mode, cfgs[3]["scan"], None)
    cnt = collections.Counter()
    for r in range(N):
        f = reg.make_matcher(mode, y, is_localization_mode=True); f.AddCloudToLocalMap([cfgs[0]["map"]])
        T = np.eye(4); ok = f.Match(cl, T, update_map=False)
        cnt[(ok, f.stats.iterations, f.stats.n_valid, f.stats.n_source, f.map_size(0), hashlib.sha1(T.tobytes()).hexdigest()[:12])] += 1
        f.close()
    print("icp fresh-handle Match", dict(cnt), flush=True)
