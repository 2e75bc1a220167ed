"""CPU-only checks of the drop-in boundary: the C-ABI library was built in-tree, loads, and exports
every function include/pies_hip.h declares (no compute call is made here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pies_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pies_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_list_the_same_symbols():
    from pies_amd import capi
    assert declared_functions() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from pies_amd import capi
    assert os.path.exists(capi.LIB_PATH), "run `python -m pies_amd.build` (the driver's build() does)"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.pies_abi_version.restype = ctypes.c_int
    assert lib.pies_abi_version() == 4


def test_options_struct_matches_reference_layout():
    from pies_amd import capi
    o = capi.Options()
    assert ctypes.sizeof(o) == 14 * 4
    lib = capi.load()
    d = capi.Options(iterations=99)
    lib.pies_default_options(ctypes.byref(d))
    got = [getattr(d, f) for f, _ in capi.Options._fields_]
    ref = [0.012, 1, 4, 4, 0.1, 0.05, 10.0, 0.006, 0.01, 0.0, 0.0, 2.0, 8, 1]  # Solver.h:23-38
    assert all(abs(a - b) < 1e-7 for a, b in zip(got, ref))


def test_no_cpu_fallback_without_device():
    """Without a gfx950 device the product refuses to create a solver (it never computes on the host)."""
    import pytest
    from pies_amd import capi
    h = ctypes.c_void_p()
    rc = capi.load().pies_create(None, 0, ctypes.byref(h))
    if rc == 0:  # running on a GPU box
        capi.load().pies_destroy(h)
        pytest.skip("a gfx950 device is present")
    assert rc == capi.ERR_HIP and not h.value
