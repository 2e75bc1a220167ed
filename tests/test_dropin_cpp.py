"""The C++ drop-in boundary: a host program written against the reference's public API
(Include/Pies/Solver.h:21-116) compiles against include/Pies/Solver.h and links libpies_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_example.cpp")
LIBDIR = os.path.join(ROOT, "pies_amd", "lib")


def build_example(tmp_path):
    exe = str(tmp_path / "dropin_example")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lpies_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def test_host_program_compiles_and_links(tmp_path):
    assert os.path.exists(build_example(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pbd", "pd"])
def test_host_program_runs(tmp_path, mode):
    out = subprocess.run([build_example(tmp_path), mode], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "dropin ok" in out.stdout
