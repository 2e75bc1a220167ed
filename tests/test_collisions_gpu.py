"""GPU parity of the PBD node-node collision pass (Src/Solver.cpp:81-130, SpatialHash.h, NodeCompRange).

The reference loop is order dependent.  The device has three orders (PIES_FLAG_COLLISION_ORDER = the oracle's rule number):
  * rule 0 - the reference's own loop (ascending node index, cell range from the node's current position), run as one
    sequential chain; the default under PIES_SCHEDULE_EXACT, compared with the oracle's plain loop (rule 0);
  * rule 2 - the pair order of DESIGN.md "Node-node collisions" (the default otherwise): the same visits, pair by pair in
    ascending order of a 64-bit mix of the two indices, executed by dependency levels; the oracle replays it with a sort
    and a sequential loop over ALL pairs that share a cell (the device lists only pairs near enough to touch and checks
    that this cannot have changed the result);
  * rule 1 - rounds 1-2's group order (27 residue classes of minimum cells), which the oracle replays as well.
In all of them positions and velocities are expected to agree bit for bit.  Gate: 1e-5 * spacing."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-5


particles = scenes.loose_particles


TURNS = "turns"  # rule 0 executed by dependency levels of turns whatever the scene's size (the default from 1 024 nodes on)


def pair(pies, oracle, build, iterations, ticks, rule=2, oracle_rule=None, **opt):
    """rule: the device's node-node order (0 the reference's - as one sequential chain below 1 024 nodes -, TURNS the reference's by
    dependency levels of turns, 1 the group order, 2 the pair order); the oracle runs the same rule (TURNS: its plain loop, rule 0)"""
    turns = rule == TURNS
    if turns:
        rule = 0
    pies.set_tuning("PIES_REFERENCE_TURNS", "1" if turns else ("0" if rule == 0 and opt.pop("chain", False) else None))
    try:
        g = pies.Solver(scenes.pbd_options(pies, iterations, **opt))
        o = oracle.OracleSolver(scenes.pbd_options(oracle, iterations, **opt))
        for s in (g, o):
            build(s)
        o.set_flag(oracle.FLAG_COLLISION_RULE, rule if oracle_rule is None else oracle_rule)
        g.set_flag(pies.FLAG_COLLISION_ORDER, rule)
        g.tick(ticks)
        o.tick(ticks)
    finally:
        pies.set_tuning("PIES_REFERENCE_TURNS", None)
    return g, o


RULES = pytest.mark.parametrize("rule", [0, TURNS, 1, 2], ids=["reference-order", "reference-order-by-turns", "group-order", "pair-order"])


def check(g, o, exact=True):
    for name in ("positions", "velocities"):
        a, b = getattr(g, name), getattr(o, name)
        assert np.isfinite(a).all()
        d = np.abs(a - b).max()
        assert d <= TOL, (name, d)
        if exact:
            assert np.array_equal(a, b), (name, d)
    assert not g.failed


@RULES
def test_two_spheres(pies, oracle, rule):
    def build(s):
        s.addNodes([[0.0, 5, 0], [0.8, 5, 0], [7.3, 5, 1.0]])
    g, o = pair(pies, oracle, build, 2, 3, rule=rule, gravity=0.0)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 0


@RULES
def test_jittered_particles_small(pies, oracle, rule):
    """BASELINE config 4 in miniature: loose particles (addNodes: radius 0.5, mass 1) on a 0.9 lattice."""
    p, v = particles((6, 7, 8))

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 4, 5, rule=rule)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1000


@RULES
def test_negative_coordinates_and_mixed_radii(pies, oracle, rule):
    rng = np.random.default_rng(5)
    p = rng.uniform(-6, 6, (900, 3)).astype(np.float32) + np.float32([0, 8, 0])
    r = rng.uniform(0.2, 0.5, 900).astype(np.float32)
    im = rng.uniform(0.5, 2.0, 900).astype(np.float32)
    v = rng.uniform(-2, 2, (900, 3)).astype(np.float32)

    def build(s):
        s.add_nodes_raw(p, vel=v, radius=r, invMass=im)
    g, o = pair(pies, oracle, build, 3, 4, rule=rule, friction=0.2, staticFrictionThreshold=0.5)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 0


def test_beam_with_constraints_and_collisions(pies, oracle):
    """The reference's default PBD tick: constraints + node-node pass + floor, every iteration.  Schedule EXACT is the
    reference's order throughout - container-order sweeps AND the ascending-index collision loop - against the oracle
    with nothing replayed; schedule COLOURED runs the exported sweep order and the parallel collision order."""
    def build(s):
        scenes.build_beam(s, (5, 4, 6), translation=(0, 0.6, 0), w_tet=0.002)
        s.set_radii(np.full(120, 0.5, np.float32))  # lattice neighbours touch; the perturbation makes them overlap
        scenes.perturb(s, 3, 0.05)
    for schedule in (0, 1):
        g = pies.Solver(scenes.pbd_options(pies, 4))
        o = oracle.OracleSolver(scenes.pbd_options(oracle, 4))
        for s in (g, o):
            build(s)
        g.set_schedule(schedule)
        if schedule == 1:
            for t in (pies.DISTANCE, pies.TET):
                o.permute(t, g.order(t))
        o.set_flag(oracle.FLAG_COLLISION_RULE, 0 if schedule == 0 else 2)
        g.tick(4); o.tick(4)
        check(g, o)
        assert g.collision_pairs == o.collision_pairs > 0
        if schedule == 0:
            assert g.launch_counts()["collide"] == 4  # the reference's loop: one launch per iteration
        else:
            assert g.collision_health()["passes_inexact"] == 0


def test_config1_lattice_reference_order(pies, oracle):
    """BASELINE config 1 (10x10x10 beam, 10 iterations) as the reference runs it by default: node-node pass on, radius
    0.475 (neighbours 0.05 apart from touching; the perturbation makes many overlap), schedule EXACT = the reference's
    order.  The oracle runs its plain loops."""
    g = pies.Solver(scenes.pbd_options(pies, 10))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 10))
    for s in (g, o):
        scenes.build_beam(s, scenes.L1K, w_tet=0.002, translation=(0, 0.5, 0))
        scenes.perturb(s, 12, 0.06)
    g.set_schedule(pies.SCHEDULE_EXACT)
    g.tick(2); o.tick(2)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1000


def test_device_rule_vs_reference_order_is_a_small_perturbation(oracle):
    """Not a gate on the product: documents how far the device's visiting rule is from the reference's
    ascending-index loop on the same scene (both orders are valid Gauss-Seidel sweeps)."""
    p, v = particles((6, 7, 8))
    res = []
    for rule in (0, 2):
        o = oracle.OracleSolver(scenes.pbd_options(oracle, 4))
        o.addNodes(p); o.set_velocities(v)
        o.set_flag(oracle.FLAG_COLLISION_RULE, rule)
        o.tick(1)
        res.append(o.positions)
    d = np.abs(res[0] - res[1]).max()
    com = np.abs(res[0].mean(0) - res[1].mean(0)).max()
    ext = np.abs((res[0].max(0) - res[0].min(0)) - (res[1].max(0) - res[1].min(0))).max()
    print("device rule vs reference order after 1 tick: max |dpos| %.3g, centre of mass %.3g, extent %.3g" % (d, com, ext))
    # an over-packed particle block relaxes violently and the sweep order matters node by node; the bulk
    # (centre of mass, extent) is what stays comparable
    assert com < 0.05 and ext < 0.5


@pytest.mark.parametrize("rule", [0, TURNS, 1, 2])
@pytest.mark.parametrize("spacing", [1.0, 0.3, 0.045])
def test_small_grid_spacing(pies, oracle, spacing, rule):
    """gridSpacing < 2 (r + 0.5): a node spans 3 and more cells per axis (NodeCompRange allows up to 50,
    Solver.cpp:896-898; at spacing 0.045 the range is 45 cells wide for r = 0.5 and EMPTY for the larger radius, which
    then never collides as a visiting node).  The group order needs ranges of at most 2 cells, so it runs the reference's
    loop in these scenes (the oracle is asked for rule 0 then); the pair order lists node by node over the node's own range,
    with the number of shared cells - up to 45^3 here - beside the entry."""
    p, v = particles((4, 3, 5), spacing=0.8)
    r = np.full(len(p), 0.5, np.float32)
    r[::7] = 0.7

    def build(s):
        s.add_nodes_raw(p, vel=v, radius=r, invMass=np.ones(len(p), np.float32))
    g, o = pair(pies, oracle, build, 2, 2, rule=rule, gridSpacing=spacing, oracle_rule=0 if rule in (1, TURNS) else rule)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 0
    if rule == 2:
        h = g.collision_health()
        assert h["levels"] > 0 and h["passes_inexact"] == 0, h


def test_wide_ranges_in_the_pair_order_at_scale(pies, oracle):
    """30 000 loose particles on a grid of 0.6 (ranges of 3-4 cells per axis, 27-64 cells per node, a pair shares up to 36 of
    them): the pair order against the oracle, two ticks."""
    p, v = particles((30, 25, 40))

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 3, 2, rule=2, gridSpacing=0.6)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 100_000
    h = g.collision_health()
    print("wide ranges, pair order:", h)
    assert h["passes_inexact"] == 0


@RULES
def test_grid_sort_follows_a_growing_box(pies, oracle, rule):
    """The node grid's radix sort is captured with as many passes as the scene's cell box needs (11 key bits per pass at most, five
    bits to spare); the host follows the box at every synchronisation.  A cluster flying apart - its box grows from 5 x 6 x 7 cells
    to sixteen times that per axis over twelve ticks, the key from 9 to 21 bits - stays exact against the oracle tick by tick."""
    p, v = particles((6, 7, 8))
    c = p.mean(0)
    v = (110.0 * (p - c) + v).astype(np.float32)     # radial: the outermost particles leave at ~350 units per second (4 per tick)

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 2, 0, rule=rule, floorHeight=-1000.0)
    for t in range(12):
        g.tick(); o.tick()
        check(g, o)
    assert np.ptp(g.positions, axis=0).max() > 8 * np.ptp(p, axis=0).max()


@RULES
def test_clusters_far_apart_take_the_unpacked_sort(pies, oracle, rule):
    """A cell key of at most 32 bits travels through the grid's sort in one word with its value; two clusters 800 000 units apart
    on every axis make a box of 57 key bits - six passes of ten bits over (key, value) pairs, the path every other test's scene
    is too small for.  Exact against the oracle in all three orders."""
    p, v = particles((5, 4, 6))
    q = (p + np.float32(8.0e5)).astype(np.float32)
    pos = np.concatenate([p, q]).astype(np.float32)
    vel = np.concatenate([v, v]).astype(np.float32)

    def build(s):
        s.addNodes(pos)
        s.set_velocities(vel)
    g, o = pair(pies, oracle, build, 2, 2, rule=rule)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 100


def test_a_box_that_outgrows_the_captured_sort_is_latched(pies):
    """...and a box that grows faster than the spare bits allow between two looks of the host (here: 60 times per axis inside the
    first tick) is a failure latch, not a wrong grid."""
    p, v = particles((6, 7, 8))
    v = (9000.0 * (p - p.mean(0))).astype(np.float32)
    g = pies.Solver(scenes.pbd_options(pies, 2, floorHeight=-1.0e5))
    g.addNodes(p)
    g.set_velocities(v)
    g.tick(2)
    assert g.failed and "cell box" in g.last_error(), g.last_error()


def test_config4_l500k_one_tick(pies, oracle):
    """BASELINE config 4 at full size (50x100x100 loose particles, 4 iterations): one tick vs the oracle."""
    p, v = particles(scenes.L500K)

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 4, 1)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1_000_000
    h = g.collision_health()
    print("config 4, pair order:", h)
    assert h["passes_inexact"] == 0 and h["levels"] > 0


def test_config4_l500k_one_iteration_reference_order(pies, oracle):
    """BASELINE config 4 at full size in the REFERENCE's order (rule 0: ascending node index, range from the node's live position),
    one iteration - 500 000 turns, >18 M resolved pairs - against the oracle's plain loop.  Exact equality.  (Until round 4 the
    device ran this as one dependent chain on one wavefront, ~20 us per node: 15 s; since round 5 by dependency levels of turns.)"""
    p, v = particles(scenes.L500K)

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 1, 1, rule=0)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 10_000_000
    # (the first iteration of the over-packed block throws nodes further than the lists of the turns cover - excursions of 0.6 -: the
    # turns notice and the sequential loop runs the pass; the settled state below runs by turns alone)
    print("config 4, first iteration in the reference order:", g.collision_health(), "passes of the sequential loop:", g.collision_fallbacks)


def test_config4_l500k_one_tick_reference_order_settled(pies, oracle):
    """A whole tick of BASELINE config 4 (four iterations) in the reference's order once the burst of the over-packed block is over
    (the state after 14 ticks of the default order): by turns alone - no pass repeated, none left to the sequential loop - against
    the oracle's plain loops.  Exact equality."""
    p, v = particles(scenes.L500K)
    w = pies.Solver(scenes.pbd_options(pies, 4))
    w.addNodes(p)
    w.set_velocities(v)
    w.tick(14)
    assert not w.failed
    p, v = w.positions, w.velocities
    w.close()

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 4, 1, rule=0)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1_000_000
    h = g.collision_health()
    print("config 4 (settled), one tick in the reference order by turns:", h, "passes of the sequential loop:", g.collision_fallbacks)
    assert h["levels"] > 100 and h["passes_inexact"] == 0 and g.collision_fallbacks == 0


def test_the_sequential_chain_still_runs_the_reference_order(pies, oracle):
    """k_collide_reference - the turns' fallback, and the path below 1 024 nodes - on a scene large enough to take the turns by default"""
    p, v = particles((12, 10, 11))

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 3, 2, rule=0, chain=True)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1000
    assert g.collision_health()["levels"] == 0


def test_dense_cells_take_the_unstaged_path(pies, oracle):
    """More distinct nodes around one cell than the resolve kernel stages in LDS (256): those groups are resolved
    straight from global memory, the others from LDS, inside the same passes; both must replay the same order."""
    rng = np.random.default_rng(7)
    dense = rng.uniform(0.05, 3.95, (700, 3)) + [0, 1.0, 0]       # ~90 nodes per 2^3 cell, ~700 around the middle ones
    loose, _ = particles((5, 5, 5), y0=1.0)
    p = np.concatenate([dense, loose + [8.0, 0, 0]]).astype(np.float32)
    r = np.concatenate([np.full(len(dense), 0.12), np.full(len(loose), 0.5)]).astype(np.float32)
    v = rng.uniform(-1, 1, p.shape).astype(np.float32)

    def build(s):
        s.add_nodes_raw(p, vel=v, radius=r, invMass=np.ones(len(p), np.float32))
    for rule in (2, 1):
        g, o = pair(pies, oracle, build, 3, 2, rule=rule)
        check(g, o)
        assert g.collision_pairs == o.collision_pairs > 700
    g, o = pair(pies, oracle, build, 2, 1, rule=0)  # and the sequential chain over the same dense buckets
    check(g, o)
    g, o = pair(pies, oracle, build, 2, 1, rule=TURNS)  # the reference order by turns: partner lists of several batches of 64
    check(g, o)


@pytest.mark.parametrize("rule", [0, TURNS, 1, 2], ids=["reference-order", "reference-order-by-turns", "group-order", "pair-order"])
@pytest.mark.parametrize("radius", [0.01, 0.5], ids=["sparse-pile", "overlapping-pile"])
def test_piles_are_left_to_the_sequential_loop(pies, oracle, rule, radius):
    """More nodes in one grid cell than any table of the parallel orders holds - until round 4 a device-only failure latch ("more than
    2 048 nodes in a cell"), which the reference's loop does not have (Solver.cpp:81-130).  2 300 nodes inside one cell:
    * tiny spheres (nobody within reach of anybody): the pair order and the turns run as usual, the group order (tables of 2 048 nodes
      per cell) hands the pass to the sequential loop;
    * spheres of radius 0.5 (everybody overlaps everybody: 2 299 partners per node, the lists hold 1 024): every parallel order
      hands the pass over.
    A pass the sequential loop ran has the REFERENCE's order whatever order was asked for: the oracle's plain loop (rule 0), bit
    for bit.  (The pair order's own result where it runs: rule 2.)"""
    rng = np.random.default_rng(3)
    p = (rng.uniform(0.6, 1.4, (2300, 3)) + [0, 4, 0]).astype(np.float32)  # all inside the cell (0, 2, 0) of the 2.0 grid
    v = rng.uniform(-1, 1, p.shape).astype(np.float32)

    def build(s):
        s.add_nodes_raw(p, vel=v, radius=radius, invMass=np.ones(len(p), np.float32))
    falls_back = radius > 0.1 or rule == 1
    g, o = pair(pies, oracle, build, 1, 1, rule=rule, oracle_rule=0 if falls_back or rule == TURNS else rule)
    check(g, o)
    pairs = g.collision_pairs  # (reading the counter clears it)
    assert pairs == o.collision_pairs
    if radius > 0.1:
        assert pairs > 100_000
        if rule in (TURNS, 2):
            assert g.collision_fallbacks >= 1


def test_resolve_variants_agree(pies, monkeypatch, tune):
    """(group order) The one-launch form resolves all 27 residue classes in one launch (tickets + completion stamps, LDS staging);
    PIES_COLLIDE_PASSES=1 is the 27-launch form and PIES_COLLIDE_GLOBAL=1 the unstaged one.  The order of
    conflicting groups is the same in all of them, so the results must be identical bit for bit."""
    p, v = particles((9, 8, 10))

    def run():
        g = pies.Solver(scenes.pbd_options(pies, 3))
        g.addNodes(p)
        g.set_velocities(v)
        g.set_flag(pies.FLAG_COLLISION_ORDER, pies.COLLISION_ORDER_GROUPS)
        g.tick(3)
        return g.positions, g.velocities, g.collision_pairs, g.launch_counts()["collide"]
    ref = run()
    assert ref[3] == 6  # per iteration: the flow kernel + the sequential loop's launch that returns at once unless a pile needs it (round 5)
    for name in ("PIES_COLLIDE_PASSES", "PIES_COLLIDE_GLOBAL"):
        tune(name, "1")
        alt = run()
        tune(name, None)
        assert np.array_equal(ref[0], alt[0]) and np.array_equal(ref[1], alt[1]) and ref[2] == alt[2], name
        assert alt[3] == (84 if name == "PIES_COLLIDE_PASSES" else 6)


def test_pair_level_variants_agree(pies, monkeypatch, tune):
    """(pair order) The level launches exist in several shapes - one lane or four per pair, workgroups of 64 / 128 / 256 threads, one
    to four of their wavefronts looking at frontier nodes with the taken pairs sorted by their visits across the workgroup, the
    repeat's levels as one launch or as captured launches, few workgroups (several turns of the loop each).  They run the same
    visits in the same order of conflicting pairs: identical results, bit for bit (30 000 particles: a level's frontier is tens of
    thousands of nodes in the first rounds and a handful in the last)."""
    p, v = particles((30, 25, 40), jitter=0.08)

    def run():
        g = pies.Solver(scenes.pbd_options(pies, 3))
        g.addNodes(p)
        g.set_velocities(v)
        g.tick(2)
        assert not g.failed
        return g.positions, g.velocities, g.collision_pairs
    ref = run()
    assert ref[2] > 50_000
    variants = [{"PIES_PAIR_QUADS": "0"}, {"PIES_PAIR_LOOK_WAVES": "1"}, {"PIES_PAIR_LOOK_WAVES": "2"},
                {"PIES_PAIR_QUAD_THREADS": "64"}, {"PIES_PAIR_QUAD_THREADS": "128", "PIES_PAIR_LOOK_WAVES": "2"},
                {"PIES_PAIR_QUAD_BLOCKS": "7"}, {"PIES_PAIR_QUAD_BLOCKS": "3", "PIES_PAIR_LOOK_WAVES": "4"},
                {"PIES_PAIR_REPEAT_LAUNCHES": "1"}]
    for var in variants:
        for name, value in var.items():
            tune(name, value)
        alt = run()
        for name in var:
            tune(name, None)
        assert np.array_equal(ref[0], alt[0]) and np.array_equal(ref[1], alt[1]) and ref[2] == alt[2], var


@RULES
def test_distant_clusters_use_wide_keys(pies, oracle, rule):
    """The sort key packs the cell coordinates relative to the bounding box of all ranges: two clusters 300 000 apart on
    every axis (plus single strays) make the box 150 000 cells per axis - 54 key bits, seven of the eight radix passes -
    where a compact scene like config 4 needs three."""
    a, va = particles((5, 4, 6), seed=7)
    b, vb = particles((4, 5, 3), seed=8)
    strays = np.float32([[-250000.3, 40.0, 7.0], [12.0, 260000.7, -3.0], [1.0, 2.0, -270000.1]])
    p = np.concatenate([a, b + np.float32([300000.0, 300000.0, 300000.0]), strays]).astype(np.float32)
    v = np.concatenate([va, vb, np.zeros_like(strays)]).astype(np.float32)

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    g, o = pair(pies, oracle, build, 3, 3, rule=rule, gravity=0.0)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 500


@pytest.mark.parametrize("spacing", [0.55, 0.7])
def test_deep_overlaps_leave_the_slack(pies, oracle, spacing):
    """The pair order lists only pairs near enough to touch while every node stays within its slack of the position the grid
    was built from.  Here overlaps are deep (spacing 0.55 / 0.7 against r = 0.5): nodes are pushed further than their slack
    allows for, the device tests the unlisted neighbours of those nodes with the excursions of the pass and, where a pair may
    have been missed, repeats the pass with those nodes listing everything around them.  The result is still the oracle's,
    which never filters."""
    p, v = particles((6, 5, 7), spacing=spacing, jitter=0.05)

    def build(s):
        s.addNodes(p)
        s.set_velocities(v)
    for rule in (2, TURNS):
        g, o = pair(pies, oracle, build, 2, 2, rule=rule)
        h = g.collision_health()
        print("deep overlaps, spacing", spacing, rule, h)
        assert h["passes_inexact"] == 0
        check(g, o)
        assert g.collision_pairs == o.collision_pairs > 1000


def test_pair_order_is_deterministic_with_levels_in_the_tail(pies):
    """Two solvers on the same input, most levels in the single-workgroup tail kernel: bit-identical positions and the same
    number of resolved pairs (a node whose partner's lane has already moved it on in this level must not be taken again)."""
    p, v = particles((30, 30, 30))
    runs = []
    for _ in range(2):
        g = pies.Solver(scenes.pbd_options(pies, 4))
        g.addNodes(p)
        g.set_velocities(v)
        g.set_collision_rounds(20)
        g.tick(3)
        runs.append((g.positions, g.velocities, g.collision_pairs))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2] > 100000


def test_pair_order_deeper_than_the_captured_rounds(pies, oracle):
    """Only three level launches are captured (pies_set_collision_rounds): the tail kernel finishes the remaining levels."""
    p, v = particles((7, 6, 8))
    g = pies.Solver(scenes.pbd_options(pies, 3))
    o = oracle.OracleSolver(scenes.pbd_options(oracle, 3))
    for s in (g, o):
        s.addNodes(p)
        s.set_velocities(v)
    o.set_flag(oracle.FLAG_COLLISION_RULE, 2)
    g.set_collision_rounds(3)
    g.tick(3); o.tick(3)
    check(g, o)
    assert g.collision_pairs == o.collision_pairs > 1000 and g.collision_health()["levels"] > 3
