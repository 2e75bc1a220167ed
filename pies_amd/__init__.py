"""pies_amd -- MI355X (gfx950) implementation of the Pies soft-body solver loop.

The product is `pies_amd/lib/libpies_hip.so` (hand-written HIP kernels behind the C ABI declared in
`include/pies_hip.h`) plus the C++ drop-in class `include/Pies/Solver.h`.  This Python package only
builds the library (`pies_amd.build`) and binds the C ABI with ctypes (`pies_amd.capi`) for the tests
and the benchmark; there is no CPU code path here.
"""
from . import capi  # noqa: F401
from .capi import Options, PiesError, Solver  # noqa: F401

__all__ = ["capi", "Options", "PiesError", "Solver"]
