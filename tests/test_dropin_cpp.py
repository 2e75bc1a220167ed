"""The C++ drop-in boundary: a host program written against the reference's public API
(Include/Pies/Solver.h:21-116) compiles against include/Pies/Solver.h and links libpies_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_example.cpp")
LIBDIR = os.path.join(ROOT, "pies_amd", "lib")


def build_example(tmp_path, src=SRC):
    exe = str(tmp_path / os.path.splitext(os.path.basename(src))[0])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
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


TETMESH = os.path.join(ROOT, "tests", "cpp", "tetmesh_example.cpp")


def test_tetmesh_program_compiles_and_links(tmp_path):
    assert os.path.exists(build_example(tmp_path, TETMESH))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["faces", "derive"])
def test_add_tet_mesh_volume_through_the_class(tmp_path, mode, oracle):
    """N4: Solver::addTriMeshVolume after its tetgen call (PrimitiveUtilities.cpp:243-328) - boundary filter, winding switch
    (:263-266), mass = density, radius 0.5, one strain + one volume constraint per element - through the C++ class.  The
    program checks counts, masses and that every surface triangle is wound outward; the positions it prints after 5 PD
    ticks (triangle contacts and floor contacts depend on the triangle list) are compared with the oracle fed the same
    mesh and the triangles the program reports."""
    import numpy as np
    from test_pd_parity_gpu import pd_options, tol_for
    out = subprocess.run([build_example(tmp_path, TETMESH), mode], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout[-400:], out.stderr)
    assert "tetmesh ok" in out.stdout
    tris = np.array([l.split()[1:] for l in out.stdout.splitlines() if l.startswith("tri ")], dtype=np.uint32)
    tets = np.array([l.split()[1:] for l in out.stdout.splitlines() if l.startswith("tet ")], dtype=np.uint32)
    pos = np.array([[float.fromhex(v) for v in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("pos ")], dtype=np.float32)
    assert len(tris) == 48 and len(tets) == 48 and len(pos) == 28
    o = oracle.OracleSolver(pd_options(oracle, 5))
    o.addNodes([[9.0, 3.0, 9.0]])
    p = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1).reshape(-1, 3) + [0.3, 1.5, 0.2]
    o.add_nodes_raw(p.astype(np.float32), vel=np.tile(np.float32([0, -1, 0]), (27, 1)), radius=0.5, invMass=np.float32(1.0) / np.float32(2.5))
    o.add_tet(tets, 1.0, 0.8, 1.0)
    o.add_volume(tets, 1.0, 1.0, 1.0)
    o.add_triangles(tris)
    o.tick(5)
    assert np.abs(pos - o.positions).max() <= tol_for(o.positions)
    assert pos[1:, 1].mean() < 2.5 - 0.05  # it fell
