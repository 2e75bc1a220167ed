"""GPU parity of the PD point-triangle collision pipeline (K3 / H3 / D1): triangle grid + swept-range
candidate search (Src/Solver.cpp:680-875), CCD (Src/CollisionDetection.cpp:227-302), contact constraint,
stabilisation and friction (Src/CollisionConstraint.cpp:67-194, Src/Solver.cpp:367-383, 431-471).

Detection decisions are taken with the same fp32 arithmetic on both sides, so from identical input state
the contact LISTS must be identical (same contacts, same order, duplicates included).  A contact decision is
discontinuous (a point on a triangle edge flips with a 1e-7 change), and the global solve carries the PD
tolerance (CG vs direct solve), so every tick is started from the oracle's state ("teacher forcing") and
compared after one tick: lists exactly, positions within 1e-5 x bounding-box diagonal."""
import numpy as np
import pytest

import scenes
from test_pd_parity_gpu import pd_options, tol_for, within, yardstick

pytestmark = pytest.mark.gpu


def sync_state(g, o):
    g.set_positions(o.positions)
    g.set_prev_positions(o.prev_positions)
    g.set_velocities(o.velocities)


def two_boxes(s, gap=0.04, vy=-2.0, offset=(0.4, 0.3)):
    s.create_tet_box(3, 3, 3, translation=(0, 0.02, 0), w=1.0)                       # sits on the floor
    s.create_tet_box(3, 3, 3, translation=(offset[0], 2.02 + gap, offset[1]), w=1.0)  # just above it, falling
    v = s.velocities
    v[27:, 1] = vy
    s.set_velocities(v)
    s.set_prev_positions(s.positions)


@pytest.mark.parametrize("contact_rows", ["inline", "pass"])
def test_two_boxes_contact_lists_and_positions(pies, oracle, monkeypatch, contact_rows, tune):
    """Both graph variants of the global step: contact rows summed by the row's lane inside the SpMV (few contacts) or
    by a wavefront per node in a pass of their own (the variant the host switches to at 512 contacts)."""
    tune("PIES_TRI_FAST_ROWS", "1" if contact_rows == "pass" else "0")
    g = pies.Solver(pd_options(pies, 6))
    o = oracle.OracleSolver(pd_options(oracle, 6))
    for s in (g, o):
        two_boxes(s)
    tol = tol_for(o.positions)
    seen = 0
    for t in range(12):
        sync_state(g, o)
        g.tick(); o.tick()
        cg_, co = g.tri_collisions, o.tri_collisions
        assert np.array_equal(cg_, co), (t, len(cg_), len(co))
        seen += len(co)
        assert np.abs(g.positions - o.positions).max() <= tol, t
        assert np.abs(g.prev_positions - o.prev_positions).max() <= tol, t
        assert np.abs(g.velocities - o.velocities).max() <= tol / 0.012, t
    assert seen > 100 and not g.failed
    # the contacting node rides on the lower box instead of falling through it (see the oracle test)
    assert g.positions[36, 1] > g.positions[:27, 1].max() - 0.1


def test_switching_the_pipeline_off_matches_oracle_too(pies, oracle):
    g = pies.Solver(pd_options(pies, 4))
    o = oracle.OracleSolver(pd_options(oracle, 4))
    for s in (g, o):
        two_boxes(s)
    g.set_flag(pies.FLAG_TRIANGLE_COLLISIONS, 0)
    o.set_flag(oracle.FLAG_TRIANGLE_COLLISIONS, 0)
    g.tick(8); o.tick(8)
    assert len(g.tri_collisions) == 0
    assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions)


def test_friction_and_static_threshold(pies, oracle):
    """sliding contact: lateral velocity, non-default friction and static threshold"""
    kw = dict(friction=0.3, staticFrictionThreshold=0.4, collisionStabilizationIterations=3)
    g = pies.Solver(pd_options(pies, 5, **kw))
    o = oracle.OracleSolver(pd_options(oracle, 5, **kw))
    for s in (g, o):
        two_boxes(s, gap=0.03, vy=-1.0, offset=(0.2, 0.1))
        v = s.velocities
        v[27:, 0] = 1.5
        s.set_velocities(v)
    tol = tol_for(o.positions)
    for t in range(10):
        sync_state(g, o)
        g.tick(); o.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions), t
        assert np.abs(g.positions - o.positions).max() <= tol, t
        assert np.abs(g.velocities - o.velocities).max() <= tol / 0.012, t


def test_many_contacts_spanning_several_windows(pies, oracle):
    """a 6x6 plate pressed onto a 6x6 plate: several hundred contacts per tick (> 64: multiple windows of the
    sequential stabilisation / friction passes, with conflicts inside and across windows)"""
    g = pies.Solver(pd_options(pies, 4))
    o = oracle.OracleSolver(pd_options(oracle, 4))
    for s in (g, o):
        s.create_tet_box(6, 2, 6, translation=(0, 0.02, 0), w=1.0)
        s.create_tet_box(6, 2, 6, translation=(0.37, 1.05, 0.41), w=1.0)
        v = s.velocities
        v[72:, 1] = -1.5
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    tol = tol_for(o.positions)
    most = 0
    for t in range(6):
        sync_state(g, o)
        g.tick(); o.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions), t
        most = max(most, len(o.tri_collisions))
        within("many_contacts", g, o, tol)
    assert most > 128


def test_sequential_passes_on_the_lds_copy_equal_the_l2_path_bit_for_bit(pies, monkeypatch, tune):
    """The stabilisation and friction passes are order dependent; run level by level they give the sequential result whatever
    the level partition.  Two device solvers on the same scene (thousands of contacts): one with the node-owner levels + the
    four-lanes-per-contact passes on an LDS copy of the touched nodes, one with the chunked levels + one lane per contact
    through L2 (PIES_TRI_LDS=0).  Everything else is the same code, so the states must agree bit for bit."""
    def run(lds):
        tune("PIES_TRI_LDS", lds)
        tune("PIES_TRI_FAST_ROWS", "1")
        g = pies.Solver(pd_options(pies, 3))
        g.set_pcg(3e-7, 64)
        g.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
        g.create_tet_box(12, 2, 18, translation=(0.37, 1.05, 0.41), w=1.0)
        v = g.velocities
        v[14 * 2 * 20:, 1] = -1.5
        g.set_velocities(v)
        g.set_prev_positions(g.positions)
        out = []
        for _ in range(4):
            g.tick()
            out.append((g.positions.copy(), g.velocities.copy(), g.prev_positions.copy(), g.tri_collisions.copy()))
        assert not g.failed
        g.close()
        return out
    a, b = run("1"), run("0")
    assert max(len(t[3]) for t in a) > 1024
    for t, (x, y) in enumerate(zip(a, b)):
        for k in range(4):
            assert np.array_equal(x[k], y[k]), (t, k)


def test_two_captured_iterations_and_the_rest_in_the_last_launch(pies, oracle, monkeypatch, tune):
    """PIES_PCG_BUDGET=2 pins the captured CG iterations at two; thousands of w = 1e4 contacts need 8-14.  The last captured
    launch goes on by itself (k_cg_update: grid barriers, contact rows summed lane by lane) and the substeps still meet the
    tolerance and stay with the oracle's direct solves."""
    tune("PIES_PCG_BUDGET", "2")
    g = pies.Solver(pd_options(pies, 3))
    o = oracle.OracleSolver(pd_options(oracle, 3))
    for s in (g, o):
        s.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
        s.create_tet_box(12, 2, 18, translation=(0.37, 1.05, 0.41), w=1.0)
        v = s.velocities
        v[14 * 2 * 20:, 1] = -1.5
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    tol = 2.0 * tol_for(o.positions)
    most = 0
    for t in range(3):
        sync_state(g, o)
        g.tick(); o.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions), t
        res, iters_used, solves = g.pcg_stats()
        most = max(most, iters_used)
        assert g.pcg_health()["budget"] == 2 and res <= 3e-7 * 1.0001, (t, res, iters_used)
        assert np.abs(g.positions - o.positions).max() <= tol, t
    assert most > 4 and g.pcg_health()["short_solves"] == 0 and not g.failed


def test_patch_wider_than_the_lds_copy_takes_the_l2_path(pies, oracle, monkeypatch, tune):
    """Two 52x52 plates: more than 4096 nodes take part in contacts, so the sequential passes cannot run on an LDS copy and
    the level kernel takes the chunked relaxation (the automatic choice, no switch set).  One teacher-forced tick against
    the oracle, and the same tick with the L2 path forced (PIES_TRI_LDS=0) must give the same bits."""
    tune("PIES_TRI_FAST_ROWS", "1")
    def build(s):
        s.create_tet_box(52, 2, 52, translation=(0, 0.02, 0), w=1.0)
        s.create_tet_box(52, 2, 52, translation=(0.37, 1.05, 0.41), w=1.0)
        v = s.velocities
        v[52 * 2 * 52:, 1] = -1.5
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    o = oracle.OracleSolver(pd_options(oracle, 3))
    build(o)
    o.tick()
    states = []
    for lds in ("1", "0"):
        tune("PIES_TRI_LDS", lds)
        g = pies.Solver(pd_options(pies, 3))
        g.set_pcg(3e-7, 256)
        build(g)
        g.tick()
        assert not g.failed
        states.append((g.positions.copy(), g.velocities.copy(), g.tri_collisions.copy()))
        g.close()
    touched = len(np.unique(o.tri_collisions))
    assert touched > 4096, touched
    assert np.array_equal(states[0][2], o.tri_collisions)
    assert np.abs(states[0][0] - o.positions).max() <= 2.0 * tol_for(o.positions)
    for a, b in zip(states[0], states[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("contact_rows,sequential", [("inline", "lds"), ("pass", "lds"), ("pass", "l2"), ("pass-unmerged", "lds")])
def test_thousands_of_contacts_level_schedule(pies, oracle, monkeypatch, contact_rows, sequential, tune):
    """A plate resting on a larger one: 2000+ contacts per tick, chains of tens of contacts through one node.  Exercises
    the dependency levels of the whole contact list (node-owner rounds; with PIES_TRI_LDS=0 the chunked relaxation, more
    than one 1024-contact chunk), the level-by-level stabilisation / friction passes on the LDS copy of the touched nodes
    and through L2, and both variants of the contact rows in the global step."""
    tune("PIES_TRI_FAST_ROWS", "0" if contact_rows == "inline" else "1")
    if contact_rows == "pass-unmerged":  # rows with more than 4 distinct columns keep the contact-by-contact form
        tune("PIES_ROW_MAX_UNIQUE", "4")
    tune("PIES_TRI_LDS", "1" if sequential == "lds" else "0")
    g = pies.Solver(pd_options(pies, 3))
    g.set_pcg(3e-7, 256)  # thousands of w = 1e4 contacts: the default cap of 32 CG iterations stops above the tolerance
    o = oracle.OracleSolver(pd_options(oracle, 3))
    for s in (g, o):
        s.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
        s.create_tet_box(12, 2, 18, translation=(0.37, 1.05, 0.41), w=1.0)
        v = s.velocities
        v[14 * 2 * 20:, 1] = -1.5
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    # 5 000+ contacts of weight 1e4 against elastic terms of order 1: the fp32 solution of that system (direct in the
    # oracle, CG here) is only good to a few 1e-5 of the body size, so twice the usual PD tolerance (measured: 1.3x with
    # the contact rows summed lane by lane, 0.5x with the pairwise sums of the wavefront pass)
    # the gate is the yardstick's (no fitted constant): the oracle with its global solve in double, teacher-forced like the device
    o64 = oracle.OracleSolver(pd_options(oracle, 3))
    o64.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
    o64.create_tet_box(12, 2, 18, translation=(0.37, 1.05, 0.41), w=1.0)
    o64.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    most = 0
    for t in range(3):
        sync_state(g, o)
        o64.set_positions(o.positions); o64.set_prev_positions(o.prev_positions); o64.set_velocities(o.velocities)
        g.tick(); o.tick(); o64.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions) and np.array_equal(o64.tri_collisions, o.tri_collisions), t
        most = max(most, len(o.tri_collisions))
        yardstick("thousands_of_contacts[%s,%s]" % (contact_rows, sequential), g, o, o64, names=("positions", "velocities"))
    assert most > 1024 and not g.failed


def test_config5_l250k_with_binding_contacts(pies, oracle):
    """BASELINE config 5, one GPU's share: a 250 000-particle body (25x25x400 lattice beam, strain + volume constraints,
    PD, 10 local/global iterations) with the point-triangle pipeline on AND binding: the beam lies on the floor (floor
    contacts along its whole length) and a second body lands on it.  Teacher-forced ticks against the oracle (direct fp32
    solve, re-ordered and re-factored with the contact blocks every substep like Solver.cpp:242-262): contact lists
    equal entry for entry, positions within the PD tolerance."""
    g = pies.Solver(pd_options(pies, 10))
    o = oracle.OracleSolver(pd_options(oracle, 10))
    W, H, D = scenes.L250K
    for s in (g, o):
        s.create_tet_box(W, H, D, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
        s.create_tet_box(8, 6, 30, translation=(3.3, 0.04 + (H - 1) + 0.04, 40.4), w=1.0, volume=True, triangles=True)
        v = s.velocities
        v[W * H * D:, 1] = -2.0
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    assert g.count(pies.NODES) == 250000 + 8 * 6 * 30
    # the yardstick (no fitted gate): the oracle's global solve in double, teacher-forced with the device every tick
    o64 = oracle.OracleSolver(pd_options(oracle, 10))
    o64.create_tet_box(W, H, D, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    o64.create_tet_box(8, 6, 30, translation=(3.3, 0.04 + (H - 1) + 0.04, 40.4), w=1.0, volume=True, triangles=True)
    o64.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    seen = 0
    for t in range(3):
        sync_state(g, o)
        o64.set_positions(o.positions); o64.set_prev_positions(o.prev_positions); o64.set_velocities(o.velocities)
        g.tick(); o.tick(); o64.tick()
        assert np.array_equal(o64.tri_collisions, o.tri_collisions)
        yardstick("config5_l250k_contacts", g, o, o64, names=("positions", "velocities"))
        cg_, co = g.tri_collisions, o.tri_collisions
        assert np.array_equal(cg_, co), (t, len(cg_), len(co))
        seen = max(seen, len(co))
        res, iters_used, solves = g.pcg_stats()
        assert solves == 10 and res <= 3e-7 * 1.0001, (t, res)
    assert seen > 200 and o.count(oracle.STATICS) > 10000 and not g.failed
    assert g.pcg_health()["short_solves"] == 0


def test_wide_triangles_use_the_reference_range_limits(pies, oracle):
    """A triangle's box over position and previous position may span up to 50 world-unit cells per axis when it is
    inserted (TriCompRange, Solver.cpp:974-976) and up to 20 when it searches (sweptTriRange, :672-674); longer ranges
    are empty.  Scene: a 30-cell-wide triangle (inserted, does not search), a 12-cell-wide one (does both), a 60-cell
    wide one (neither), and a small tet box falling through them."""
    big = np.float32([[-5.2, 1.30, -5.1], [24.7, 1.32, -5.3], [-5.4, 1.31, 24.6],       # 30 cells in x and z
                      [-2.3, 0.90, -2.2], [9.4, 0.93, -2.1], [-2.2, 0.91, 9.5],         # 12 cells
                      [-30.5, 0.5, -30.2], [29.6, 0.52, -30.1], [-30.3, 0.51, 29.4]])   # 60 cells: empty range
    g = pies.Solver(pd_options(pies, 5))
    o = oracle.OracleSolver(pd_options(oracle, 5))
    for s in (g, o):
        s.addNodes(big)
        s.add_triangles([[0, 1, 2], [3, 4, 5], [6, 7, 8]])
        s.create_tet_box(3, 3, 3, translation=(0.35, 1.40, 0.45), scale=0.6, w=1.0)
        v = s.velocities
        v[9:, 1] = -6.0
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
    # the PD tolerance is relative to the extent of the system that is solved: the CG stops on the residual relative to the
    # whole right-hand side, which the corners of the wide triangles (coordinates up to 30) dominate, and the contacts couple
    # the falling body to them (measured on the body's nodes: 4.1e-5, its own bounding box would allow 4.08e-5)
    tol_scene = tol_for(o.positions)
    seen = 0
    for t in range(8):
        sync_state(g, o)
        g.tick(); o.tick()
        cg_, co = g.tri_collisions, o.tri_collisions
        assert np.array_equal(cg_, co), (t, len(cg_), len(co))
        seen += len(co)
        assert np.abs(g.positions - o.positions).max() <= tol_scene, t
    assert seen > 20 and not g.failed and not o.failed
    hit = set(int(b) for b in np.unique(o.tri_collisions[:, 1:])) if len(o.tri_collisions) else set()
    assert not (hit & {6, 7, 8})  # nothing ever touches the triangle whose range is empty


def test_few_triangles_spanning_many_cells(pies, oracle):
    """with 2 triangles the list of used cells held 64 x 2 words while one 30 x 30-cell triangle creates
    ~1 800 cells - a write past the array.  Storage now has a floor (2^18 entries) and the append is bounded: the scene runs
    like the oracle's."""
    big = np.float32([[-5.2, 1.30, -5.1], [24.7, 1.32, -5.3], [-5.4, 1.31, 24.6],   # 30 cells in x and z
                      [-2.3, 0.90, -2.2], [9.4, 0.93, -2.1], [-2.2, 0.91, 9.5]])    # 12 cells
    g = pies.Solver(pd_options(pies, 4))
    o = oracle.OracleSolver(pd_options(oracle, 4))
    for s in (g, o):
        s.addNodes(big)
        s.add_triangles([[0, 1, 2], [3, 4, 5]])
        s.set_prev_positions(s.positions)
    tol = tol_for(o.positions)
    for t in range(4):
        sync_state(g, o)
        g.tick(); o.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions), t
        assert np.abs(g.positions - o.positions).max() <= tol, t
    assert not g.failed and not o.failed


def test_triangles_of_fifty_cells_per_axis_run_like_the_reference(pies, oracle):
    """Three slanted triangles spanning 50 cells on every axis: the reference lists each in 125 000 cells (TriCompRange accepts
    50 cells per axis, Solver.cpp:974-976) and searches none of them (sweptTriRange stops at 20, :672-674).  Until round 4 the
    device's grid held (cell, triangle) entries like the reference's and latched a failure when 375 000 of them met the 2^18
    reserved; a triangle is now listed once (tri_kernels.h), so the scene runs, without contacts, like the oracle's."""
    c = np.float32([[0.2, 0.3, 0.1], [49.6, 49.5, 0.4], [0.3, 49.7, 49.5]])
    nodes = np.concatenate([c + np.float32([60.0, 0.0, 0.0]) * k for k in range(3)])
    g = pies.Solver(pd_options(pies, 2))
    o = oracle.OracleSolver(pd_options(oracle, 2))
    for s in (g, o):
        s.addNodes(nodes)
        s.add_triangles([[0, 1, 2], [3, 4, 5], [6, 7, 8]])
        s.set_prev_positions(s.positions)
    tol = tol_for(o.positions)
    for t in range(3):
        sync_state(g, o)
        g.tick(); o.tick()
        assert np.array_equal(g.tri_collisions, o.tri_collisions), t
        assert np.abs(g.positions - o.positions).max() <= tol, t
    assert not g.failed and not o.failed


def _soup(seed, n_tri=360):
    """Random triangles of very different sizes in a box of 20 x 10 x 20 world units (edge lengths 0.3-12, three of 25-45: listed
    but too long to search), every triangle with nodes of its own."""
    rng = np.random.default_rng(seed)
    size = np.exp(rng.uniform(np.log(0.3), np.log(9.0), n_tri))
    size[:3] = rng.uniform(25.0, 45.0, 3)
    centre = rng.uniform(-10.0, 10.0, (n_tri, 3)) * [1.0, 0.5, 1.0] + [0.0, 30.0, 0.0]      # well above the floor
    # (flat in y: a range of more than 1000 cells that is short enough to search fails the sim, Solver.cpp:741-745)
    corners = centre[:, None, :] + rng.uniform(-0.5, 0.5, (n_tri, 3, 3)) * size[:, None, None] * [1.0, 0.3, 1.0]
    for _ in range(8):  # (a triangle at rest must not trip that latch either: halve the ones that would, jitter included)
        length = np.ceil(corners.max(1) + 0.3) - np.floor(corners.min(1) - 0.3)
        heavy = (length <= 23).all(1) & (length.prod(1) > 800)
        corners[heavy] = centre[heavy, None, :] + 0.5 * (corners[heavy] - centre[heavy, None, :])
    return corners.reshape(-1, 3).astype(np.float32), np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_triangle_soup_contact_lists_equal_the_oracles(pies, oracle, seed):
    run_soup(pies, oracle, seed)


def test_triangle_soup_in_the_contact_heavy_graph_variant(pies, oracle, tune):
    """The same soup with the graph variant for many contacts pinned (list offsets, list and incidence chain as launches of their
    own - k_tri_scan, k_tri_emit over the whole grid, k_inc_* - instead of the single-workgroup k_tri_tail)."""
    tune("PIES_TRI_FAST_ROWS", "1")
    run_soup(pies, oracle, 15)


def run_soup(pies, oracle, seed):
    """The broad phase lists every triangle once (minimum-corner cell, three size classes, slots modulo the table: tri_kernels.h)
    where the reference lists it in every cell of its range; the contact list must come out the same, entry for entry, duplicates
    included.  Triangle soup: sizes over two orders of magnitude (all three classes in use), random velocities up to 12 units per
    tick (swept ranges of every length up to the search limit and beyond), a third of the triangles carried 1 000 and 2 000
    units away after finalize (their cells share the table's slots with the others'), a new random state every tick."""
    nodes, tris = _soup(seed)
    g = pies.Solver(pd_options(pies, 1))
    o = oracle.OracleSolver(pd_options(oracle, 1))
    for s in (g, o):
        s.addNodes(nodes)
        s.add_triangles(tris)
        s.set_prev_positions(s.positions)
    rng = np.random.default_rng(1000 + seed)
    h = 0.012                                    # SolverOptions::fixedTimestepSize, one substep per tick
    total, classes = 0, np.zeros(3, dtype=np.int64)
    for t in range(5):
        p = nodes + rng.uniform(-0.3, 0.3, nodes.shape).astype(np.float32)
        far = (np.arange(len(nodes)) // 3) % 3                       # per triangle: 0 stays, 1 and 2 are carried away
        p[:, 0] += np.float32(1000.0) * far
        p[:, 2] -= np.float32(500.0) * (far == 2)
        step = rng.uniform(-1.0, 1.0, nodes.shape) * rng.choice([0.05, 1.0, 12.0], (len(nodes) // 3, 1)).repeat(3, 0)
        v = (step / h).astype(np.float32)
        # a triangle whose swept range is short enough to search (20 cells per axis) but holds more than 1000 cells would trip the
        # reference's latch: it stays where it is in this tick
        q = p + np.float32(h) * v
        both = np.concatenate([p.reshape(-1, 3, 3), q.reshape(-1, 3, 3)], axis=1)
        length = np.ceil(both.max(1)) - np.floor(both.min(1))
        heavy = (length <= 20).all(1) & (length.prod(1) > 900)
        v[np.repeat(heavy, 3)] = 0.0
        for s in (g, o):
            s.set_positions(p)
            s.set_prev_positions(p)
            s.set_velocities(v)
        g.tick(); o.tick()
        assert not o.failed and not g.failed, (t, g.last_error())
        cg_, co = g.tri_collisions, o.tri_collisions
        assert np.array_equal(cg_, co), (t, len(cg_), len(co))
        total += len(co)
        st = g.tri_grid_stats()
        classes += np.asarray(st["listed"])
    assert total > 300, total
    assert (classes > 0).all(), classes      # all three size classes held triangles
    return total


def test_a_crowd_below_the_reference_latch_is_no_failure(pies, oracle):
    """Solver.cpp:751-755 fails the sim when a bucket holds more than 1000 triangles.  The device has no buckets: a triangle whose
    search windows list more than 1000 triangles counts, cell by cell of its range, the ranges that hold the cell.  Two clusters of
    600 small triangles in neighbouring cells: every window lists 1 200, no cell holds more than 600 - no failure on either side,
    the same contacts; with 1 050 in one of the cells both sides latch."""
    for crowd, fails in ((600, False), (1050, True)):
        rng = np.random.default_rng(5)
        a = rng.uniform(0.10, 0.90, (3 * crowd, 3)) + [0.0, 5.0, 0.0]
        b = rng.uniform(0.10, 0.90, (3 * 600, 3)) + [1.0, 5.0, 0.0]
        # small triangles: every corner within 0.05 of the first one (a range of one cell)
        for c in (a, b):
            c[1::3] = c[0::3] + rng.uniform(-0.04, 0.04, (len(c) // 3, 3))
            c[2::3] = c[0::3] + rng.uniform(-0.04, 0.04, (len(c) // 3, 3))
        nodes = np.concatenate([a, b]).astype(np.float32)
        g = pies.Solver(pd_options(pies, 1))
        o = oracle.OracleSolver(pd_options(oracle, 1))
        for s in (g, o):
            s.addNodes(nodes)
            s.add_triangles(np.arange(len(nodes), dtype=np.uint32).reshape(-1, 3))
            s.set_prev_positions(s.positions)
        g.tick(); o.tick()
        assert o.failed == fails and g.failed == fails, (crowd, o.failed, g.failed)
        if not fails:
            assert g.tri_grid_stats()["listed"][0] == crowd + 600
            assert np.array_equal(g.tri_collisions, o.tri_collisions)
