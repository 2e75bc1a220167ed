import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "benchlib")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api
    oracle_api.lib()
    return oracle_api


@pytest.fixture(scope="session")
def pies():
    """The product binding.  GPU tests must run on the HIP library: no fallback, fail loudly."""
    from pies_amd import capi
    capi.load()
    return capi


@pytest.fixture
def tune(pies):
    """tune(name, value): a tuning / diagnostic switch of the library for the duration of one test (pies_set_tuning; value None
    unsets).  These switches are not read from the environment."""
    touched = []

    def _set(name, value):
        touched.append(name)
        pies.set_tuning(name, value)
    yield _set
    for name in touched:
        pies.set_tuning(name, None)


@pytest.fixture(scope="session", autouse=True)
def bounds_report():
    """With the bounds-checking build loaded (PIES_LIB=.../libpies_hip_bounds.so, python -m pies_amd.build --bounds): after the
    session, no kernel may have recorded an out-of-range index (dev_math.h PIES_IN_BOUNDS)."""
    yield
    if "bounds" not in os.environ.get("PIES_LIB", ""):
        return
    import ctypes
    from pies_amd import capi
    L = capi.load()
    out = (ctypes.c_uint * 6)()
    L.pies_exp_bounds_report.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    assert L.pies_exp_bounds_report(out) == 0
    print("\n[bounds build] violations (first site, count) layer %s pd %s cg1 %s" % (tuple(out[0:2]), tuple(out[2:4]), tuple(out[4:6])))
    assert not any(out), tuple(out)
