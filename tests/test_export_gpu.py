"""Render-state export (Solver.h:42-71; Solver.cpp:157,393 write _vertices every substep): one pinned D2H copy per
pies_tick, and the double-buffered asynchronous form (pies_tick_begin / pies_export_acquire / pies_export_release) in
which frame k travels to the host while frame k+1 computes."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _beam(pies, solver=None, iterations=4):
    g = pies.Solver(scenes.pbd_options(pies, iterations) if solver is None else solver)
    scenes.build_beam(g, (5, 5, 12))
    scenes.perturb(g, 2, 0.05)
    g.set_flag(pies.FLAG_NODE_COLLISIONS, 0)
    return g


def test_async_frames_equal_synchronous_ticks(pies):
    a, b = _beam(pies), _beam(pies)
    ref = []
    for _ in range(6):
        a.tick()
        ref.append(a.positions.copy())
    frames = []
    f1 = b.tick_begin()
    f2 = b.tick_begin()          # two frames in flight
    assert (f1, f2) == (1, 2)
    for k in range(6):
        f = k + 1
        view = b.export_acquire(f)
        assert view.shape == (300, 4)
        frames.append(view[:, :3].copy())
        assert np.array_equal(view[:, 3], b.inv_masses)  # the record's fourth float is invMass
        b.export_release(f)
        if f + 2 <= 6:
            assert b.tick_begin() == f + 2
    for k in range(6):
        assert np.array_equal(frames[k], ref[k]), k
    # the state in HBM is the last frame's
    assert np.array_equal(b.positions, ref[-1])


def test_third_begin_while_oldest_frame_is_held_is_refused(pies):
    g = _beam(pies)
    f1 = g.tick_begin()
    g.export_acquire(f1)         # held
    g.tick_begin()
    with pytest.raises(pies.PiesError):
        g.tick_begin()           # would overwrite the buffer of the held frame
    g.export_release(f1)
    g.tick_begin()
    with pytest.raises(pies.PiesError):
        g.export_acquire(f1)     # only the last two frames are kept


def test_tick_leaves_positions_current_and_strided_read_matches(pies):
    g = _beam(pies)
    g.tick(3)
    p = g.positions
    v = g.read_positions_strided(9)  # Solver::Vertex is 9 floats (36 bytes), position first
    assert np.array_equal(v[:, :3], p) and not v[:, 3:].any()
    # the other arrays are fetched on demand, one array per request
    assert np.isfinite(g.velocities).all() and np.isfinite(g.prev_positions).all()
    q = g.positions
    assert np.array_equal(p, q)


def test_pd_export_and_release_hinge_recapture(pies, oracle):
    """PD through the asynchronous path, and a flag flip between ticks (re-capture without re-upload)."""
    from test_pd_parity_gpu import build_pd_beam, pd_options, tol_for
    g = pies.Solver(pd_options(pies, 5))
    o = oracle.OracleSolver(pd_options(oracle, 5))
    for s in (g, o):
        build_pd_beam(s, (4, 4, 8))
        scenes.perturb(s, 9, 0.03)
        s.set_prev_positions(s.positions)
    for k in range(4):
        f = g.tick_begin()
        o.tick()
        view = g.export_acquire(f)
        assert np.abs(view[:, :3] - o.positions).max() <= tol_for(o.positions)
        g.export_release(f)
    g.set_flag(pies.FLAG_RELEASE_HINGE, 1)   # PD ignores the hinge (Solver.cpp:59 is PBD only): same results
    g.tick(); o.tick()
    assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions)
