"""Product host logic (C++ in funny_lidar_slam_amd/csrc/host_maps.hpp) exercised on the CPU: compiled with hipcc,
no GPU needed (no HIP runtime call is made).  See tests/host/host_logic_test.cpp for what is checked."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_bookkeeping(built, tmp_path):
    exe = os.path.join(str(tmp_path), "host_logic_test")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result",
           os.path.join(ROOT, "tests", "host", "host_logic_test.cpp"), "-o", exe, "-L" + os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host logic ok" in out.stdout, out.stdout + out.stderr


def test_exact_sort_closed_form_model(tmp_path):
    """The partition closed form the device sort is built on (tests/host/exact_sort_model_test.cpp) reproduces std::sort's permutation."""
    exe = os.path.join(str(tmp_path), "exact_sort_model_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "exact_sort_model_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_eviction_selection_with_recreated_voxels_model(tmp_path):
    """The batch form of 'evict inside the insert loop' (ivox_map.cpp:133-136) incl. voxels evicted and re-created by one batch -- the walk
    ivox_evict_select performs -- equals the sequential loop on random maps (tests/host/evict_conflict_model_test.cpp)."""
    exe = os.path.join(str(tmp_path), "evict_conflict_model_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "evict_conflict_model_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
