"""Both forms of the normal-equation reduction stay under test: the default build sums J J^T through the FP64
matrix cores (v_mfma_f64_16x16x4_f64 on v v^T, measured faster), `-DFLS_FIT_MFMA=0` (libfls_reg_dpp.so, csrc/Makefile) builds the DPP
row-shift tree north_star's wording asks for ("wavefront shuffle reductions ... no MFMA").  The parity tests of the two kinds that use the
reduction run once more against that library, in a process of their own (the library is chosen at import: FLS_REG_LIB)."""
import os
import subprocess
import sys

import pytest

from funny_lidar_slam_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dpp_tree_build_passes_the_parity_tests(built):
    assert _lib.device_count() >= 1
    lib = os.path.join(ROOT, "funny_lidar_slam_amd", "libfls_reg_dpp.so")
    assert os.path.exists(lib), "csrc/Makefile builds libfls_reg_dpp.so next to libfls_reg.so"
    env = dict(os.environ, FLS_REG_LIB=lib)
    sel = "(test_config2_p2plane_ivox and 0.05) or (test_config4_loam_full and 0.1) or test_determinism_two_runs_bit_identical"
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", sel],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    tail = run.stdout.strip().splitlines()[-1]
    assert " passed" in tail and "failed" not in tail, tail
    n_passed = int(tail.split(" passed")[0].split()[-1])
    assert n_passed >= 3, tail
    print("DPP-tree build:", tail)
