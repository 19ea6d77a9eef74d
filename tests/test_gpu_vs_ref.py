"""GPU <-> compiled reference, DIRECTLY, both behind std::shared_ptr<RegistrationInterface> (VERDICT r2 missing #5 / next #1b).

oracle/_ref/gpu_vs_ref (tests/harness/gpu_vs_ref.cpp, built by oracle/ref_shim/Makefile where /root/reference exists; the
binary travels to the GPU box) holds the reference's own class -- compiled verbatim from /root/reference -- and the product's
HipRegistration adapter -- compiled against the reference's REAL registration_interface.h / data_type.h /
pointcloud_cluster.h -- in one program and replays the scenarios of tests/refpin.py (mapping mode: ivox, ivox_lru, icp,
ndt, ndt_dev, loam; localization mode + GetFitnessScore: icp_loc, kd_loc, ivox_loc, ndt_loc) through both with the
pipeline's call sequence (src/slam/frontend.cpp:125-140,208; localization.cpp:135-138), one fresh process per scenario
(SURVEY Q12).  Per frame: same return value, pose within BASELINE's 1e-4 m / 1e-4 rad (observed: <= 1e-9), fitness equal.

Until now the chain was transitive (GPU <-> oracle, oracle <-> compiled reference); this closes it.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import refpin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "gpu_vs_ref")
need_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/gpu_vs_ref absent and /root/reference not present to build it")

KINDS = {"IcpOptimized": 0, "PointToPlane_IVOX": 1, "IncrementalNDT": 2, "LoamFull_KdTree": 3, "PointToPlane_KdTree": 4}


def _cloud_bytes(c):
    if c is None:
        return struct.pack("<Q", 0)
    a = np.asarray(c, np.float32)
    out = np.zeros((a.shape[0], 4), np.float32)
    out[:, :3] = a[:, :3]
    if a.shape[1] >= 8:
        out[:, 3] = a[:, 4]
    elif a.shape[1] >= 4:
        out[:, 3] = a[:, 3]
    return struct.pack("<Q", out.shape[0]) + out.tobytes()


def write_scenario(path, sc, params):
    """FLSSCN1 file of tests/harness/gpu_vs_ref.cpp::load"""
    import ctypes as C
    with open(path, "wb") as f:
        f.write(b"FLSSCN1\0")
        f.write(struct.pack("<4i", KINDS[sc["mode"]], int(sc["loc"]), len(sc["init_clouds"]), len(sc["frames"])))
        f.write(struct.pack("<q", int(sc.get("ivox_capacity", 0))))
        f.write(struct.pack("<I", C.sizeof(params)))
        f.write(bytes(params))
        for c in sc["init_clouds"]:
            f.write(_cloud_bytes(c))
        for fr in sc["frames"]:
            f.write(_cloud_bytes(fr["scan"]))
            f.write(_cloud_bytes(fr["corner"]))
            absolute = "absolute_guess" in fr
            step = fr["absolute_guess"] if absolute else fr["guess_step"]
            f.write(np.ascontiguousarray(np.asarray(step, np.float64).T).tobytes())  # column-major
            f.write(struct.pack("<i", int(absolute)))


def parse_out(path):
    frames, summary = [], None
    for line in open(path):
        t = line.split()
        if t[0] == "frame":
            d = {t[i]: t[i + 1] for i in range(2, len(t), 2)}
            frames.append(dict(ref_ok=int(d["ref_ok"]), hip_ok=int(d["hip_ok"]), dt=float(d["dt"]), dr=float(d["dr"]),
                               ref_fitness=float(d["ref_fitness"]), hip_fitness=float(d["hip_fitness"]), ref_ms=float(d["ref_ms"]), hip_ms=float(d["hip_ms"])))
        elif t[0] in ("ref_T", "hip_T"):
            frames[-1][t[0]] = np.array([float(x) for x in t[1:]]).reshape(4, 4).T
        elif t[0] == "summary":
            summary = {t[i]: float(t[i + 1]) for i in range(1, len(t), 2)}
    return frames, summary


def params_for(sc):
    """the fls_params the Python mirror builds for this mode string + YAML block (same struct the C++ side reads)"""
    from funny_lidar_slam_amd import registration as reg
    m = reg.make_matcher(sc["mode"], sc["y"], is_localization_mode=sc["loc"])
    p = m.params
    m.close()
    return p


@need_exe
def test_harness_compiled_against_the_real_reference_headers():
    out = subprocess.run([EXE, "--compile-check"], capture_output=True, text=True)
    assert out.returncode == 0 and "real headers" in out.stdout


@need_exe
@pytest.mark.gpu
@pytest.mark.parametrize("name", refpin.SCENARIOS)
def test_hip_registration_equals_the_compiled_reference_through_the_interface(built, tmp_path, name):
    sc = refpin.make_scenario(name)
    scn, outp = os.path.join(str(tmp_path), "scn.bin"), os.path.join(str(tmp_path), "out.txt")
    write_scenario(scn, sc, params_for(sc))
    run = subprocess.run([EXE, scn, outp], capture_output=True, text=True, timeout=900)
    assert run.returncode in (0, 1), run.stdout + run.stderr
    frames, summary = parse_out(outp)
    assert len(frames) == len(sc["frames"]) and summary is not None
    for k, fr in enumerate(frames):
        assert fr["ref_ok"] == fr["hip_ok"], (name, k, fr)
        assert fr["dt"] <= 1e-4 and fr["dr"] <= 1e-4, (name, k, fr["dt"], fr["dr"])
        # observed agreement is far inside the contract: every scenario must stay below 1e-8 (FP64 through differently associated
        # but equivalent arithmetic; a flipped correspondence or keyframe decision would show up as >= 1e-6)
        assert fr["dt"] <= 1e-8 and fr["dr"] <= 1e-8, (name, k, fr["dt"], fr["dr"])
        if sc["loc"]:
            a, b = fr["ref_fitness"], fr["hip_fitness"]
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (name, k, a, b)
    assert run.returncode == 0, run.stdout + run.stderr
    print(f"{name}: {len(frames)} frames, worst |dt| {summary['worst_dt']:.2e} m, |dR| {summary['worst_dr']:.2e} rad; "
          f"reference {np.median([f['ref_ms'] for f in frames]):.1f} ms vs HIP {np.median([f['hip_ms'] for f in frames]):.2f} ms per Match (median)")
