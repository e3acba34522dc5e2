"""composable_kernel_amd -- MI355X (gfx950) weight-quantized GEMM engine.

The product is ``csrc/libwqgemm.so`` (hand-written HIP kernels behind the C ABI declared in
``include/wqgemm.h``) plus the C++ facade under ``include/ck/``.  This Python package is the thin
ctypes plumbing the tests, ``bench.py`` and the multi-GPU split-K driver use to reach that C ABI
with torch-owned device memory.  There is no CPU fallback: importing :mod:`capi` without the
built library raises.
"""
from . import capi  # noqa: F401
from .capi import Problem, lib  # noqa: F401

__all__ = ["capi", "Problem", "lib"]
