"""funny_lidar_slam_amd -- MI355X (gfx950) back-end for funny_lidar_slam's scan-to-map registration.

Only the registration hot path lives here: the C ABI shared library
``libfls_reg.so`` built from ``csrc/`` (hand-written HIP kernels + host logic) and a thin
Python mirror of the reference's plug-in interface for tests and ``bench.py``.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "registration", "synth"]
