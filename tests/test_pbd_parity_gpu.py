"""GPU parity of the PBD substep (Src/Solver.cpp:40-160) against the CPU oracle, through the C ABI.

Tolerance: the device arithmetic is the same IEEE fp32 sequence as the oracle's (no FMA contraction,
correctly rounded sqrt/div), so positions are expected to agree bit for bit; the gate is
max|dpos| <= 1e-5 * lattice spacing after T <= 10 ticks (north_star: "a stated FP tolerance on node
positions"), and exact equality is reported (and required where no libm call is involved)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

TOL = 1e-5  # x lattice spacing (1.0)


def _pair(pies, oracle, dims, iterations, schedule, ticks, seed=7, amp=0.05, **beam):
    g = pies.Solver(scenes.pbd_options(pies, iterations))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, iterations))
    for s in (g, o):
        scenes.build_beam(s, dims, **beam)
        scenes.perturb(s, seed, amp)
        s.set_flag(1, 0)  # node-node collisions off (BASELINE configs 1-2)
    g.set_schedule(schedule)
    if schedule != pies.SCHEDULE_EXACT:  # the order the device plan is equivalent to
        for t in (pies.POSITION, pies.DISTANCE, pies.TET, pies.BEND):
            if g.count(t):
                o.permute(t, g.order(t))
    g.tick(ticks)
    o.tick(ticks)
    return g, o


def _check(g, o, exact=True):
    for name in ("positions", "velocities", "prev_positions"):
        a, b = getattr(g, name), getattr(o, name)
        assert np.isfinite(a).all()
        d = np.abs(a - b).max()
        assert d <= TOL, (name, d)
        if exact:
            assert np.array_equal(a, b), (name, d)


def test_scene_arrays_match_oracle(pies, oracle):
    g = pies.Solver(scenes.pbd_options(pies, 1))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 1))
    for s in (g, o):
        scenes.build_beam(s, (4, 5, 6), volume=True, triangles=True)
    for t in (pies.DISTANCE, pies.TET, pies.VOLUME, pies.TRIANGLES, pies.LINES):
        assert np.array_equal(g.ids(t), o.ids(t))
    for t in (pies.DISTANCE, pies.TET, pies.VOLUME):
        assert np.array_equal(g.rest(t), o.rest(t))
    assert np.array_equal(g.positions, o.positions)
    assert np.array_equal(g.radii, o.radii) and np.array_equal(g.inv_masses, o.inv_masses)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_config1_l1k_pbd(pies, oracle, schedule):
    """BASELINE config 1: 10x10x10 lattice, distance + tet-strain, 10 iterations."""
    g, o = _pair(pies, oracle, scenes.L1K, 10, schedule, ticks=5)
    _check(g, o)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_distance_only(pies, oracle, schedule):
    g, o = _pair(pies, oracle, (6, 7, 5), 4, schedule, ticks=8, tets=False)
    _check(g, o)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_tets_only_with_inversion(pies, oracle, schedule):
    # large perturbation: inverted and strongly compressed elements exercise the sigma flip and clamps
    g, o = _pair(pies, oracle, (5, 4, 6), 6, schedule, ticks=4, amp=0.8, distance=False)
    _check(g, o)


def test_exact_schedule_is_reference_order(pies, oracle):
    """EXACT never needs the oracle to replay an order: it is the container order."""
    g, o = _pair(pies, oracle, (7, 3, 9), 5, pies.SCHEDULE_EXACT, ticks=3)
    assert np.array_equal(g.positions, o.positions)


def test_exact_whole_substep_dag_equals_per_container_levels(pies, oracle, monkeypatch, tune):
    """EXACT runs one launch per level of the whole-substep dependency DAG (wavefront.cpp); PIES_NO_WAVEFRONT=1
    keeps the older one-launch-per-container-level path.  Same order of every conflicting pair, so the same bits,
    with far fewer launches; bend + position constraints and the collision pass (a barrier) included."""
    def run():
        g = pies.Solver(scenes.pbd_options(pies, 6))
        scenes.build_beam(g, (6, 5, 12))
        g.create_bend_sheet(6, 6, translation=(9.0, 3.0, 0.0))
        scenes.perturb(g, 3, 0.05)
        g.set_schedule(pies.SCHEDULE_EXACT)
        g.tick(3)
        lc = g.launch_counts()
        return g.positions, g.velocities, lc
    tune("PIES_NO_WAVEFRONT", "1")
    p0, v0, lc0 = run()
    tune("PIES_NO_WAVEFRONT", None)
    p1, v1, lc1 = run()
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    assert lc0["wave"] == 0 and lc1["wave"] > 0 and lc1["tet"] == 0
    # the collision passes cut the DAG once per iteration, so the saving is modest here (942 against 1536 launches)
    assert lc1["wave"] < lc0["distance"] + lc0["tet"] + lc0["bend"] + lc0["position"] + lc0["floor"]
    assert lc1["collide"] == lc0["collide"] == 6  # one resolve launch per iteration


def test_coloured_batches_are_conflict_free(pies):
    g = pies.Solver(scenes.pbd_options(pies, 1))
    scenes.build_beam(g, (6, 6, 6))
    g.set_flag(1, 0)
    g.set_schedule(pies.SCHEDULE_COLOURED)
    for t, writes in ((pies.DISTANCE, [0]), (pies.TET, [0, 1, 2, 3])):
        ids, order, offs = g.ids(t), g.order(t), g.batches(t)
        assert sorted(order.tolist()) == list(range(len(ids)))
        for b in range(len(offs) - 1):
            sel = ids[order[offs[b]:offs[b + 1]]]
            written = sel[:, writes].ravel()
            assert len(np.unique(written)) == len(written)
            reads = np.setdiff1d(sel.ravel(), written)
            assert len(np.intersect1d(reads, written)) == 0


def test_floor_and_friction(pies, oracle):
    # beam resting on / penetrating the floor: clamp, and the velocity pass's hard-coded speed 5.0
    g, o = _pair(pies, oracle, (5, 5, 5), 4, 0, ticks=10, translation=(0.0, 0.2, 0.0))
    _check(g, o)
    assert (g.positions[:, 1] >= g.radii - 1e-6).all()


def test_sheet_position_constraints_and_release_hinge(pies, oracle):
    for hinge in (0, 1):
        g = pies.Solver(scenes.pbd_options(pies, 6))
        g.set_schedule(pies.SCHEDULE_EXACT)  # the oracle sweeps in container order
        o = oracle.OracleSolver(scenes.pbd_options(oracle, 6))
        for s in (g, o):
            s.create_sheet(9, 7, translation=(0, 3, 0), scale=0.5, mass=2.0, w=0.7)
            scenes.perturb(s, 3, 0.05)
            s.set_flag(1, 0)
            s.set_flag(0, hinge)
        g.tick(4)
        o.tick(4)
        _check(g, o)


def test_bend_sheet(pies, oracle):
    # acosf differs between device libm and glibc by a few ulp: tolerance only
    g = pies.Solver(scenes.pbd_options(pies, 5))
    g.set_schedule(pies.SCHEDULE_EXACT)
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 5))
    for s in (g, o):
        s.create_bend_sheet(8, 8, translation=(0, 4, 0), scale=1.0, w=0.6)
        scenes.perturb(s, 5, 0.1)
        s.set_flag(1, 0)
    g.tick(5)
    o.tick(5)
    _check(g, o, exact=False)


def test_random_graph_ragged(pies, oracle):
    """Non-lattice connectivity, duplicate position constraints, batch sizes not a multiple of 64/256."""
    rng = np.random.default_rng(11)
    n = 777
    pos = rng.uniform(0, 6, size=(n, 3)).astype(np.float32) + np.float32([0, 3, 0])
    tets = np.array([rng.choice(n, 4, replace=False) for _ in range(1501)], dtype=np.uint32)
    dist = np.array([rng.choice(n, 2, replace=False) for _ in range(2003)], dtype=np.uint32)
    pins = np.array([3, 3, 10, 500, 3], dtype=np.uint32)
    im = rng.uniform(0.5, 2.0, size=n).astype(np.float32)
    for schedule in (0, 1, 2):
        g = pies.Solver(scenes.pbd_options(pies, 3))
        o = oracle.OracleSolver(scenes.pbd_options(oracle, 3))
        for s in (g, o):
            s.add_nodes_raw(pos, radius=0.1, invMass=im)
            s.add_position(pins, 0.3)
            s.add_distance(dist, 0.4)
            s.add_tet(tets, 0.02)
            s.set_flag(1, 0)
        g.set_schedule(schedule)
        if schedule != 0:
            for t in (pies.POSITION, pies.DISTANCE, pies.TET):
                o.permute(t, g.order(t))
        g.tick(3)
        o.tick(3)
        _check(g, o)


def _layered_pair(pies, oracle, build, iterations, ticks, collisions=0, hinge=0):
    g = pies.Solver(scenes.pbd_options(pies, iterations))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, iterations))
    for s in (g, o):
        build(s)
        s.set_flag(1, collisions)
        s.set_flag(0, hinge)
    if collisions:
        o.set_flag(oracle.FLAG_COLLISION_RULE, 2)  # the device's default order for the node-node pass: the pair order
    g.set_schedule(pies.SCHEDULE_LAYERED)
    g.finalize()
    for t in (pies.POSITION, pies.DISTANCE, pies.TET, pies.BEND):
        if g.count(t):
            o.permute(t, g.order(t))
    g.tick(ticks)
    o.tick(ticks)
    return g, o


def test_layered_is_active_and_fuses_an_iteration_into_two_launches(pies, oracle):
    """Schedule LAYERED on a beam: breadth-first levels are the cross-sections, one workgroup per pair of levels keeps
    the node records in LDS; an iteration is two launches, predict and velocity ride along (2*iterations + 1)."""
    def build(s):
        scenes.build_beam(s, (6, 5, 40))
        scenes.perturb(s, 3, 0.05)
    g, o = _layered_pair(pies, oracle, build, 7, ticks=4)
    lc = g.launch_counts()
    assert lc["layer"] == 2 * 7 + 1 and lc["tet"] == 0 and lc["distance"] == 0 and lc["predict"] == 0 and lc["velocity"] == 0, lc
    _check(g, o)


def test_layered_colour_classes_larger_than_the_workgroup(pies, oracle, monkeypatch, tune):
    """Cross-sections of 576 nodes give colour classes of ~270 tetrahedra and ~500 distance constraints; with the
    workgroup forced to 256 lanes every class takes two or three passes of the in-kernel loops."""
    tune("PIES_LAYER_BLOCK", "256")
    def build(s):
        scenes.build_beam(s, (24, 24, 26))
        scenes.perturb(s, 3, 0.05)
    g, o = _layered_pair(pies, oracle, build, 3, ticks=2)
    assert g.launch_counts()["layer"] == 2 * 3 + 1
    assert np.diff(g.batches(pies.TET).astype(np.int64)).max() > 256
    _check(g, o)


def test_layered_with_bend_position_constraints_and_two_bodies(pies, oracle):
    def build(s):
        scenes.build_beam(s, (4, 5, 17), translation=(0.0, 0.3, 0.0))  # touches the floor
        s.create_bend_sheet(7, 9, translation=(9.0, 3.0, 0.0))
        s.create_sheet(9, 7, translation=(20, 3, 0), scale=0.5, mass=2.0, w=0.7)  # hinged: position constraints
        s.addNodes(np.float32([[40, 0.4, 0], [41, 6, 1], [42, 0.2, 2]]))  # loose nodes ride along in some tile
        scenes.perturb(s, 4, 0.05)
    for hinge in (0, 1):
        g, o = _layered_pair(pies, oracle, build, 5, ticks=4, hinge=hinge)
        assert g.launch_counts()["layer"] > 0
        _check(g, o, exact=False)  # bend: acos


def test_layered_with_node_collisions(pies, oracle):
    """The collision pass (global launches) cuts the layer launches once per iteration."""
    def build(s):
        scenes.build_beam(s, (4, 4, 12), translation=(0.0, 2.0, 0.0))
        scenes.build_beam(s, (4, 4, 12), translation=(1.3, 5.2, 0.4))
        scenes.perturb(s, 5, 0.05)
    g, o = _layered_pair(pies, oracle, build, 4, ticks=3, collisions=1)
    lc = g.launch_counts()
    assert lc["layer"] > 0 and lc["collide"] > 0
    _check(g, o)


def test_layered_wide_body_is_cut_into_strips(pies, oracle):
    """Cross-sections of 4900 nodes, 343k particles: a pair of levels is cut into strips by a second levelling (four
    phases per container instead of two; predict / floor / velocity run as launches of their own over the level-ordered
    copy).  A squat body below 300k particles is left to the coloured schedule."""
    def small(s):
        scenes.build_beam(s, (40, 40, 40))  # 3200 nodes in a pair of levels, 64k particles
        scenes.perturb(s, 6, 0.05)
    g, o = _layered_pair(pies, oracle, small, 2, ticks=1)
    assert g.launch_counts()["layer"] == 0 and g.launch_counts()["tet"] > 0
    _check(g, o)

    def build(s):
        scenes.build_beam(s, (70, 70, 70))
        scenes.perturb(s, 6, 0.05)
    g, o = _layered_pair(pies, oracle, build, 2, ticks=1)
    lc = g.launch_counts()
    assert lc["layer"] == 2 * 7 and lc["tet"] == 0 and lc["predict"] == 1 and lc["floor"] == 2 and lc["velocity"] == 1, lc
    _check(g, o)


@pytest.mark.parametrize("candidate", [0, 1, 2])
@pytest.mark.parametrize("collisions", [0, 1])
def test_layered_strips_small_tiles_all_containers(pies, oracle, monkeypatch, collisions, tune, candidate):
    """The strip path forced onto small scenes (tiny tiles, ragged strips): beams, a bend sheet, a hinged sheet with
    position constraints (two of them on one node), with and without the collision pass between the sweeps - in the original
    plan and in round 4's two other candidates (layer_plan.cpp; a candidate that is no plan for this scene leaves the original)."""
    tune("PIES_LAYER_PLAN_FORCE", str(candidate))
    tune("PIES_LAYER_ONE_STRIP_MAX", "40")
    tune("PIES_LAYER_TILE_NODES", "90")
    tune("PIES_LAYER_STRIPS_MIN_NODES", "0")
    def build(s):
        scenes.build_beam(s, (7, 6, 13), translation=(0.0, 0.3, 0.0))
        scenes.build_beam(s, (5, 9, 4), translation=(1.3, 7.2, 0.4))
        s.create_bend_sheet(7, 9, translation=(12.0, 3.0, 0.0))
        s.create_sheet(9, 7, translation=(24, 3, 0), scale=0.5, mass=2.0, w=0.7)
        s.add_position(np.array([3, 3, 40], dtype=np.uint32), 0.3)
        s.addNodes(np.float32([[40, 0.4, 0], [41, 6, 1], [42, 0.2, 2], [43, 3, 3], [44, 0.1, 4]]))  # loose: per-node steps only
        scenes.perturb(s, 4, 0.05)
    g, o = _layered_pair(pies, oracle, build, 4, ticks=3, collisions=collisions)
    lc = g.launch_counts()
    assert lc["layer"] > 0 and lc["floor"] == 4 and lc["position"] >= 4 and (lc["collide"] > 0) == bool(collisions), lc
    _check(g, o, exact=False)  # bend: acos


def test_layered_substeps_and_state_edits(pies, oracle):
    """timeSubsteps > 1 (the captured substep is replayed) and node state written between ticks (the first launch of a
    substep reads the node array, not the level-ordered copy)."""
    g = pies.Solver(scenes.pbd_options(pies, 3, timeSubsteps=3))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 3, timeSubsteps=3))
    for s in (g, o):
        scenes.build_beam(s, (4, 4, 15))
        s.set_flag(1, 0)
    g.set_schedule(pies.SCHEDULE_LAYERED)
    g.finalize()
    for t in (pies.DISTANCE, pies.TET):
        o.permute(t, g.order(t))
    for s in (g, o):
        s.tick(2)
        scenes.perturb(s, 9, 0.1)
        s.set_radii(np.full(s.count(pies.NODES), 0.3, dtype=np.float32))
        s.tick(2)
    assert g.launch_counts()["layer"] == 7
    _check(g, o)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_unstructured_delaunay_beam(pies, oracle, schedule):
    """A tetgen-like unstructured beam (Delaunay triangulation of a jittered lattice: ~25 tetrahedra per node, irregular
    valence, no lattice colouring proposals): every schedule reproduces the oracle bit for bit."""
    mesh = scenes.delaunay_beam((7, 6, 40))
    g = pies.Solver(scenes.pbd_options(pies, 5))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 5))
    for s in (g, o):
        scenes.build_unstructured(s, mesh)
        scenes.perturb(s, 12, 0.03)
        s.set_flag(1, 0)
    g.set_schedule(schedule)
    g.finalize()
    if schedule != 0:
        for t in (pies.DISTANCE, pies.TET):
            o.permute(t, g.order(t))
    g.tick(3)
    o.tick(3)
    if schedule == 2:
        assert g.launch_counts()["layer"] == 2 * 5 + 1
    _check(g, o)


@pytest.mark.parametrize("candidate", [1, 2])
@pytest.mark.parametrize("mesh", ["delaunay", "lattice"])
def test_layered_candidate_plans(pies, oracle, tune, mesh, candidate):
    """Round 4: schedule LAYERED tries two more plans beside its original one and takes the one with the fewest colour steps
    (layer_plan.cpp): constraints whose nodes share one breadth-first level dealt to the group below or their own, whichever
    leaves their busiest node less busy (candidate 1), and the same with slabs by position as levels (candidate 2).  Each one,
    forced, reproduces the oracle replaying its order bit for bit, on an unstructured beam and on a lattice."""
    tune("PIES_LAYER_PLAN_FORCE", str(candidate))
    g = pies.Solver(scenes.pbd_options(pies, 5))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 5))
    for s in (g, o):
        if mesh == "delaunay":
            scenes.build_unstructured(s, scenes.delaunay_beam((7, 6, 40)))
        else:
            scenes.build_beam(s, (5, 4, 30))
        scenes.perturb(s, 12, 0.03)
        s.set_flag(1, 0)
    g.set_schedule(pies.SCHEDULE_LAYERED)
    g.finalize()
    orders = {t: g.order(t) for t in (pies.DISTANCE, pies.TET)}
    for t, order in orders.items():
        assert sorted(order.tolist()) == list(range(len(order)))      # a permutation of the container
        o.permute(t, order)
    g.tick(3)
    o.tick(3)
    assert g.launch_counts()["layer"] == 2 * 5 + 1
    _check(g, o)
    # the forced plan is not the original one
    tune("PIES_LAYER_PLAN_FORCE", "0")
    h = pies.Solver(scenes.pbd_options(pies, 5), device=-1)
    if mesh == "delaunay":
        scenes.build_unstructured(h, scenes.delaunay_beam((7, 6, 40)))
    else:
        scenes.build_beam(h, (5, 4, 30))
    h.set_flag(1, 0)
    h.set_schedule(pies.SCHEDULE_LAYERED)
    assert not np.array_equal(h.order(pies.TET), orders[pies.TET]) or not np.array_equal(h.order(pies.DISTANCE), orders[pies.DISTANCE])


def test_empty_and_unconstrained(pies, oracle):
    g = pies.Solver(scenes.pbd_options(pies, 2))
    g.set_schedule(pies.SCHEDULE_EXACT)
    g.set_flag(1, 0)
    g.tick()  # no nodes: no-op
    assert g.count(pies.NODES) == 0
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 2))
    p = np.float32([[0, 5, 0], [1, 0.2, 0], [2, 9, 3]])
    for s in (g, o):
        s.addNodes(p)
        s.set_flag(1, 0)
        s.tick(20)
    _check(g, o)


def test_edit_after_tick_and_substeps(pies, oracle):
    g = pies.Solver(scenes.pbd_options(pies, 3, timeSubsteps=3))
    g.set_schedule(pies.SCHEDULE_EXACT)
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 3, timeSubsteps=3))
    for s in (g, o):
        scenes.build_beam(s, (3, 3, 3))
        s.set_flag(1, 0)
        s.tick(2)
        scenes.build_beam(s, (3, 4, 3), translation=(6, 4, 0))  # second body appended after ticking
        s.tick(2)
    _check(g, o)


def test_scale_1m_layered_strips(pies, oracle):
    """The 1M-particle lattice of bench.py's scale measurements (100^3: 5.8M tetrahedra, 6.9M distance constraints) under
    schedule LAYERED with strips, two iterations, one tick: exact equality with the oracle replaying the exported order."""
    g, o = _pair(pies, oracle, scenes.L1M, 2, pies.SCHEDULE_LAYERED, ticks=1)
    lc = g.launch_counts()
    assert lc["layer"] == 2 * 7 and lc["tet"] == 0, lc
    _check(g, o)


def test_config2_l100k_one_tick_layered(pies, oracle):
    """BASELINE config 2 at full size under schedule LAYERED (the one bench.py reports): exact equality."""
    g, o = _pair(pies, oracle, scenes.L100K, 20, pies.SCHEDULE_LAYERED, ticks=1)
    _check(g, o)
    assert g.launch_counts()["layer"] == 41


def test_config2_l100k_one_tick_reference_order(pies, oracle):
    """BASELINE config 2 at full size in the REFERENCE's order (schedule EXACT: every container swept in the order the host added
    the constraints; 2 879 launches): one tick against the oracle's plain loops, nothing replayed.  Exact equality."""
    g, o = _pair(pies, oracle, scenes.L100K, 20, pies.SCHEDULE_EXACT, ticks=1)
    _check(g, o)
    assert g.launch_counts()["wave"] > 2000


def test_config2_l100k_one_tick(pies, oracle):
    """BASELINE config 2 at full size (20x20x250, 20 iterations): one tick against the oracle, coloured
    schedule replayed, plus a checksum over both schedules' launch plans."""
    g, o = _pair(pies, oracle, scenes.L100K, 20, pies.SCHEDULE_COLOURED, ticks=1)
    assert g.count(pies.TET) == 539334 and g.count(pies.DISTANCE) == 649156
    _check(g, o)
    lc = g.launch_counts()
    assert lc["tet"] == 20 * (len(g.batches(pies.TET)) - 1)


@pytest.mark.parametrize("schedule", [1, 2])
def test_release_hinge_toggled_between_ticks_recaptures_only(pies, oracle, schedule):
    """Solver::releaseHinge flipped at run time (the reference's hosts do): with per-container batches only the launch sequence
    changes, so the substep is captured again without a re-plan or re-upload - and the exported order stays what the oracle
    replays."""
    g = pies.Solver(scenes.pbd_options(pies, 5))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 5))
    for s in (g, o):
        s.create_sheet(9, 7, translation=(0, 3, 0), scale=0.5, mass=2.0, w=0.7)
        scenes.perturb(s, 3, 0.05)
        s.set_flag(1, 0)
    g.set_schedule(schedule)
    g.finalize()
    for t in (pies.POSITION, pies.DISTANCE):
        if g.count(t):
            o.permute(t, g.order(t))
    for hinge in (0, 1, 0, 1):
        for s in (g, o):
            s.set_flag(0, hinge)
            s.tick(2)
        _check(g, o)


def test_schedule_environment_override(pies, monkeypatch):
    """PIES_SCHEDULE picks the schedule new handles start with (PIES_SCHEDULE_DEFAULT = LAYERED otherwise)."""
    def launches():
        g = pies.Solver(scenes.pbd_options(pies, 4))
        scenes.build_beam(g, (6, 6, 12))
        g.set_flag(1, 0)
        lc = g.launch_counts()
        g.close()
        return lc
    assert launches()["layer"] > 0
    monkeypatch.setenv("PIES_SCHEDULE", "exact")
    lc = launches()
    assert lc["layer"] == 0 and lc["wave"] > 0
    monkeypatch.setenv("PIES_SCHEDULE", "coloured")
    lc = launches()
    assert lc["layer"] == 0 and lc["wave"] == 0 and lc["tet"] > 0
