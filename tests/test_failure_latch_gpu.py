"""The sticky failure latch (Solver.cpp:26-28: tick is a no-op once _simFailed; :741-755, :853-856 set it; clear() does
not reset it, :488-507).  Every device-side latch bit is tripped once."""
import numpy as np
import pytest

import scenes
from test_pd_parity_gpu import pd_options

pytestmark = pytest.mark.gpu


def _after_failure(g, pies):
    assert g.failed
    p = g.positions
    g.tick(2)                       # no-op
    assert np.array_equal(g.positions, p)
    g.tick_async(); g.synchronize()
    assert np.array_equal(g.positions, p)
    g.clear()                       # Solver::clear does not reset _simFailed
    g.addNodes([[0, 5, 0], [3, 5, 0]])
    q = g.positions
    g.tick()
    assert g.failed and np.array_equal(g.positions, q)


def test_a_pile_beyond_the_fallback_budget_latches(pies, tune):
    """The reference's PBD loop has no latch (Solver.cpp:81-130): a pile the parallel orders cannot run is left to the sequential
    loop - tests/test_collisions_gpu.py::test_piles_are_left_to_the_sequential_loop - unless that pass would cost more candidate tests
    than PIES_FALLBACK_VISITS allows (1e9 by default, about a minute on one wavefront): the one limit this build has and the reference
    has not.  Here the budget is set below the pile's 2 300^2."""
    tune("PIES_FALLBACK_VISITS", "1000000")
    rng = np.random.default_rng(3)
    p = (rng.uniform(0.6, 1.4, (2300, 3)) + [0, 4, 0]).astype(np.float32)  # all inside the cell (0, 2, 0) of the 2.0 grid
    g = pies.Solver(scenes.pbd_options(pies, 2))
    g.add_nodes_raw(p, radius=0.01, invMass=np.ones(len(p), np.float32))
    g.set_flag(pies.FLAG_COLLISION_ORDER, pies.COLLISION_ORDER_GROUPS)  # (the group order's tables hold 2 048 nodes of a cell)
    g.tick()
    assert "pile-up" in g.last_error()
    _after_failure(g, pies)


def test_non_finite_position_latches(pies):
    g = pies.Solver(scenes.pbd_options(pies, 2))
    p = np.float32([[0, 5, 0], [1, 5, 0], [2, 5, 0]])
    g.addNodes(p)
    g.tick()
    assert not g.failed
    p[1, 0] = np.inf
    g.set_positions(p)
    g.tick_async()
    g.synchronize()                 # an asynchronous loop learns about the failure here
    assert g.failed and "non-finite" in g.last_error()


def test_reference_order_pass_latches_too(pies):
    g = pies.Solver(scenes.pbd_options(pies, 2))
    g.set_schedule(pies.SCHEDULE_EXACT)
    p = np.float32([[0, 5, 0], [1, 5, 0], [np.nan, 5, 0]])
    g.addNodes(p)
    g.tick()
    assert g.failed


def test_more_than_1000_triangles_in_a_cell_latches(pies):
    """the reference's own safety latch (Solver.cpp:751-755)"""
    g = pies.Solver(pd_options(pies, 2))
    rng = np.random.default_rng(1)
    p = (rng.uniform(0.05, 0.95, (3300, 3)) + [0, 2, 0]).astype(np.float32)
    g.addNodes(p)
    g.add_triangles(np.arange(3300, dtype=np.uint32).reshape(-1, 3))  # 1100 small triangles inside one world-unit cell
    g.tick()
    assert "1000 triangles" in g.last_error()
    _after_failure(g, pies)


def test_collision_wait_timeout_latches(pies, monkeypatch, tune):
    """k_collide_flow's bounded wait: with a spin limit of one poll a wavefront gives up as soon as a predecessor is
    not finished yet, which latches bit 8 (a limit of 0 waits for ever; the default is ~0.3 s)."""
    tune("PIES_COLLIDE_SPIN_LIMIT", "1")
    rng = np.random.default_rng(5)
    W = 40
    p = np.stack(np.meshgrid(np.arange(W), np.arange(W), np.arange(W), indexing="ij"), -1).reshape(-1, 3) * 0.9
    p = (p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]).astype(np.float32)
    g = pies.Solver(scenes.pbd_options(pies, 4))
    g.addNodes(p)
    g.tick()
    if g.failed:                    # 64k nodes: some wavefront practically always has to wait once
        assert "timed out" in g.last_error()
    tune("PIES_COLLIDE_SPIN_LIMIT", "0")
    h = pies.Solver(scenes.pbd_options(pies, 4))
    h.addNodes(p)
    h.tick()
    assert not h.failed
