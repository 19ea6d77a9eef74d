"""Generates tests/golden/ref_*.npz from oracle/_ref/libref.so -- the reference's own registration / feature classes
compiled verbatim from /root/reference (oracle/ref_shim).  Only runs where /root/reference is present; the committed
files let the oracle be checked against the compiled reference anywhere (tests/test_ref_pin.py).
Run from the repo root:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import refpin  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for name in refpin.SCENARIOS:
        out = refpin.run_ref_subprocess(name)
        np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **refpin.to_golden(out))
        print(name, "frames", out["n_frames"], "ok", [out[k]["ok"] for k in range(out["n_frames"])], "iters", [out[k]["iters"] for k in range(out["n_frames"])])
    for name in refpin.FEATURE_CASES:
        out = refpin.run_features_subprocess(name)
        np.savez_compressed(os.path.join(HERE, f"ref_features_{name}.npz"), **{k: (np.array(refpin._sha(v)) if isinstance(v, np.ndarray) else np.array(v)) for k, v in out.items()})
        print("features", name, {k: v.shape for k, v in out.items() if isinstance(v, np.ndarray) and k in ("ordered", "corner", "planar")})
