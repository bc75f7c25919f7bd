"""dynamic_vins_amd — MI355X (gfx950) implementation of dynamic_vins' hot path
(front-end feature tracking + sliding-window bundle adjustment) behind a C ABI.

    csrc/      hand-written HIP kernels + the C ABI (include/dvins.h) -> lib/libdvins_hip.so
    host/      C++ drop-in shims with the reference's class signatures (FeatureTracker, Estimator)
    frontend.py, _abi.py   ctypes mirror used by tests/ and bench.py

There is no CPU fallback: importing works anywhere, creating a Context needs the built library
and a GPU.
"""
from ._abi import DvinsError, LIB_PATH, load  # noqa: F401

__all__ = ["DvinsError", "LIB_PATH", "load"]
