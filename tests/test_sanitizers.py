"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side: the oracle (make -C oracle asan) and the HOST
side of the product library - scene construction, the three schedulers, the LAYERED planner, the PD set-up and tile plans, the C
ABI - built by g++ with -fsanitize=address,undefined (pies_amd/build.py build_asan) and driven through host-only handles
(PIES_DEVICE_NONE).  No GPU: device code is not instrumented (GPU ASan / XNACK are not available on the pool).  The sanitizer
runtime has to be the first library of the process, so everything runs in a child python with it preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitizer_env():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc's libasan.so is not installed")
    sys.path.insert(0, ROOT)
    from pies_amd import build as b
    lib = b.build_asan()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PIES_LIB=lib, PIES_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libpies_oracle_asan.so"))
    return env


def run_child(env, args, timeout=900):
    r = subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout[-4000:] + r.stderr[-4000:]


def test_the_harness_sees_an_overflow(sanitizer_env):
    """Positive control: a deliberate out-of-bounds write through the oracle's ABI must kill the child."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, oracle_api as ora\n"
            "a = np.eye(3, dtype=np.float32); s = np.zeros(1, np.float32); b = np.zeros(9, np.float32); v = np.zeros(9, np.float32)\n"
            "ora.lib().ora_svd3(ora._pf(a), ora._pf(s), ora._pf(b), ora._pf(v))  # writes s[0..2]\n"
            "print('survived')\n") % (ROOT, os.path.join(ROOT, "benchlib"))
    rc, out = run_child(sanitizer_env, ["-c", code])
    assert rc != 0 and "heap-buffer-overflow" in out and "survived" not in out, out


def test_host_logic_and_oracle_soak_under_sanitizers(sanitizer_env):
    """Random scenes through every planner / scheduler path, PD set-up of every constraint kind, the oracle's PBD / PD /
    collision loops: no sanitizer report."""
    rc, out = run_child(sanitizer_env, [os.path.join(ROOT, "tests", "sanitizer_child.py"), "12"])
    assert rc == 0 and "sanitizer child ok" in out, out


def test_cpu_test_files_under_sanitizers(sanitizer_env):
    """The host-logic and oracle test files themselves, with both libraries instrumented."""
    rc, out = run_child(sanitizer_env, ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                                        os.path.join(ROOT, "tests", "test_host_logic.py"), os.path.join(ROOT, "tests", "test_oracle_golden.py")])
    assert rc == 0 and " passed" in out, out
