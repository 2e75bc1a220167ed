"""GPU parity of the Projective-Dynamics substep (Src/Solver.cpp:162-486) against the CPU oracle and the
fp64 golden restatement, through the C ABI.

The reference solves the global step with a sparse Cholesky factorisation (Eigen SimplicialLLT, fp32); the
device uses Jacobi-preconditioned CG, so parity is "CG converged to a relative residual" against the
oracle's direct fp32 solve.  Stated tolerance on node positions after <= 6 ticks: 1e-5 * the body's
bounding-box diagonal (i.e. 1e-4 * spacing for the 10^3 lattice): both sides carry the fp32 round-off of a
system whose right-hand side is ~ (m/h^2) |x| ~ 7e3 |x|, so the attainable accuracy scales with |x|."""
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
REL = 1e-5


def tol_for(p):
    return REL * float(np.linalg.norm(p.max(0) - p.min(0))) + 2e-5


# Every comparison with the oracle is also RECORDED: the largest |device - oracle| per test, as a multiple of the lattice spacing
# (1.0 in all these scenes) and of the gate it was held against.  The record goes to gpurun_out/pd_deviation.json (DESIGN.md
# section 7 quotes it; profiles/r06_pd_deviation.json is a copy).  The full-size scenes (config 3, config 5's body with binding
# contacts, the 5 000-contact plates) have NO fitted gate (rounds 3-5 held them against "four times the largest deviation
# measured"): they are held against the yardstick - yardstick() below - and their distance from the fp32 oracle is recorded only.
_RECORD = {}


def record(test, what, deviation, gate):
    import json
    e = _RECORD.setdefault(test, {}).setdefault(what, {"max_deviation": 0.0, "gate": gate})
    e["max_deviation"] = max(e["max_deviation"], float(deviation))
    e["gate"] = float(gate)
    e["deviation_over_gate"] = e["max_deviation"] / e["gate"] if e["gate"] else None
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "pd_deviation.json"), "w") as f:
            json.dump(_RECORD, f, indent=1, sort_keys=True)


def within(test, g, o, gate, what="positions"):
    """max |device - oracle| of one state array against `gate` (recorded, then asserted)"""
    d = float(np.abs(getattr(g, what) - getattr(o, what)).max())
    record(test, what, d, gate)
    assert np.isfinite(getattr(g, what)).all() and d <= gate, (test, what, d, gate)
    return d


def yardstick(test, g, o32, o64, spacing=1.0, names=("positions",), dt=0.012):
    """The PD tolerance against a yardstick instead of a fitted gate: o64 is the oracle with its global solve in double (the same
    fp32 matrix and right-hand side, FLAG_PD_SOLVE_FP64) - what the reference's direct solve would return without fp32 round-off.
    Per state array: |device - fp64| and |oracle32 - fp64| are recorded, and the device passes when it is no further from the fp64
    solve than twice what the reference's own fp32 arithmetic is, or within SURVEY 8c's 1e-4 x spacing (velocities: that / dt).
    |device - oracle32| is recorded without a gate of its own."""
    out = None
    for name in names:
        a, b, c = getattr(g, name), getattr(o32, name), getattr(o64, name)
        assert np.isfinite(a).all(), (test, name)
        d_dev, d_ref = float(np.abs(a - c).max()), float(np.abs(b - c).max())
        gate = max(2.0 * d_ref, 1e-4 * spacing / (dt if name == "velocities" else 1.0))
        suffix = "" if name == "positions" else "[%s]" % name
        record(test, "device_vs_fp64" + suffix, d_dev, gate)
        record(test, "oracle32_vs_fp64" + suffix, d_ref, gate)
        record(test, "device_vs_oracle32" + suffix, float(np.abs(a - b).max()), 0.0)
        assert d_dev <= gate, (test, name, "device vs fp64 %.3g, oracle32 vs fp64 %.3g, gate %.3g" % (d_dev, d_ref, gate))
        if out is None:
            out = (d_dev, d_ref)
    return out


G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pd_options(mod, iterations, **kw):
    o = dict(solver=mod.PD, iterations=iterations)
    o.update(kw)
    return mod.Options(**o)


def build_pd_beam(s, dims, pins=True, w=1.0, translation=(0.0, 0.02, 0.0)):
    W, H, D = dims
    s.create_tet_box(W, H, D, translation=translation, w=w, volume=True, triangles=True)
    if pins:  # clamp the k = 0 end cap (SURVEY config 3)
        ids = [0 + D * (j + H * i) for i in range(W) for j in range(H)]
        s.add_position(np.array(ids, dtype=np.uint32), 2.0)


def test_pd_tiny_scene_against_fp64_golden(pies):
    d = np.load(os.path.join(G, "pd_tiny.npz"))
    W, H, D = (int(v) for v in d["dims"])
    g = pies.Solver(pd_options(pies, int(d["iterations"])))
    g.create_tet_box(W, H, D, translation=d["translation"], w=1.0, volume=True, triangles=True)
    g.add_position(d["pins"], float(d["w_pin"]))
    g.set_positions(d["pos"]); g.set_prev_positions(d["pos"]); g.set_velocities(d["vel"])
    for exp in d["expected"]:
        g.tick()
        assert np.abs(g.positions - exp).max() <= 2e-4
    res, iters, solves = g.pcg_stats()
    assert solves == int(d["iterations"]) and res <= 1e-6 and iters < 12


@pytest.mark.parametrize("dims,iters", [((10, 10, 10), 10), ((4, 5, 23), 6)])
def test_pd_beam_against_oracle(pies, oracle, dims, iters):
    g = pies.Solver(pd_options(pies, iters))
    o = oracle.OracleSolver(pd_options(oracle, iters))
    o64 = oracle.OracleSolver(pd_options(oracle, iters))
    o64.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    for s in (g, o, o64):
        build_pd_beam(s, dims)
        scenes.perturb(s, 9, 0.03)
        s.set_prev_positions(s.positions)
    TOL = tol_for(o.positions)
    for t in range(5):
        g.tick(); o.tick(); o64.tick()
        for name in ("positions", "velocities", "prev_positions"):
            within("pd_beam_%dx%dx%d" % dims, g, o, TOL * (1.0 if name != "velocities" else 1.0 / 0.012), name)
        yardstick("pd_beam_%dx%dx%d" % dims, g, o, o64)
    res, iters_used, solves = g.pcg_stats()
    assert res <= 1e-6 and iters_used < 12 and solves == iters
    assert o.count(oracle.STATICS) > 0  # floor contacts active (duplicated per triangle incidence)


def test_pd_no_pins_free_fall_and_substeps(pies, oracle):
    g = pies.Solver(pd_options(pies, 4, timeSubsteps=2))
    o = oracle.OracleSolver(pd_options(oracle, 4, timeSubsteps=2))
    for s in (g, o):
        build_pd_beam(s, (3, 3, 5), pins=False, translation=(0, 2.0, 0))
        s.create_box(3, 3, 5, w=0.7, existing_offset=0, triangles=False)  # PD distance constraints too
        scenes.perturb(s, 4, 0.02)
        s.set_prev_positions(s.positions)
        s.tick(6)
    assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions)


def test_pd_rest_state_fixed_point(pies):
    g = pies.Solver(pd_options(pies, 3, gravity=0.0))
    g.create_tet_box(4, 4, 4, translation=(0, 5, 0), w=1.0)
    p0 = g.positions
    g.tick(3)
    assert np.abs(g.positions - p0).max() < 5e-5


def test_pd_stiff_system_needs_more_cg_iterations(pies, oracle):
    """w = 300: K is no longer dominated by M/h^2; the captured iteration budget is a parameter."""
    g = pies.Solver(pd_options(pies, 5))
    o = oracle.OracleSolver(pd_options(oracle, 5))
    g.set_pcg(3e-7, 64)
    for s in (g, o):
        build_pd_beam(s, (5, 5, 8), w=300.0)
        scenes.perturb(s, 2, 0.02)
        s.set_prev_positions(s.positions)
        s.tick(3)
    res, iters_used, _ = g.pcg_stats()
    assert res <= 1e-6 and 3 < iters_used < 64
    assert np.abs(g.positions - o.positions).max() <= 3 * tol_for(o.positions)


def _two_boxes(s):
    s.create_tet_box(4, 5, 7, translation=(0.0, 0.02, 0.0), w=1.0, volume=True, triangles=True)
    s.create_tet_box(2, 2, 2, translation=(7.0, 0.5, 1.0), w=3.0, volume=True, triangles=True)   # another material
    n = len(s.positions)
    s.addNodes(np.array([[0, 9, 0], [1.1, 9, 0], [0, 10.2, 0.1], [0.2, 9.1, 0.9]], np.float32))      # one more element: an odd total
    s.add_tet([n, n + 1, n + 2, n + 3], 2.0)
    s.add_volume([n, n + 1, n + 2, n + 3], 2.0)
    scenes.perturb(s, 17, 0.05)
    s.set_prev_positions(s.positions)


def test_pd_local_step_variants_agree(pies, oracle, tune):
    """The strain + volume local step exists in three forms: one element per lane (the arithmetic of round 2 but for the single
    recomposition), two elements per lane in packed fp32 with the per-element constants (PIES_PD_REST_DICT=0), and the same with
    the rest dictionary (the default; this scene has a dozen distinct sets of constants over 439 element pairs - an odd count,
    so the last lane holds one element).  All three against the oracle within the tolerance, and against each other within a
    tenth of it: they differ by roundings of the volume projection and of the recomposition only."""
    o = oracle.OracleSolver(pd_options(oracle, 8))
    _two_boxes(o)
    assert (o.count(oracle.TET) % 2) == 1
    o.tick(3)
    TOL = tol_for(o.positions)
    res = []
    for packed, rest in (("0", None), ("1", "0"), ("1", "1")):
        tune("PIES_PD_LOCAL_PACKED", packed)
        tune("PIES_PD_REST_DICT", rest)
        g = pies.Solver(pd_options(pies, 8))
        _two_boxes(g)
        g.tick(3)
        assert not g.failed and np.isfinite(g.positions).all()
        d = float(np.abs(g.positions - o.positions).max())
        record("pd_local_variants", "positions_packed%s_dict%s" % (packed, rest), d, TOL)
        assert d <= TOL, (packed, rest, d, TOL)
        res.append(g.positions)
        g.close()
    assert np.abs(res[0] - res[1]).max() <= 0.1 * TOL and np.abs(res[1] - res[2]).max() <= 0.1 * TOL


def test_row_dictionary_changes_nothing(pies, tune):
    """The system matrix as a row dictionary (rows with the same stencil share one copy: column offsets from the row and values)
    against the SELL arrays: the same entries in the same order, so the same sums - positions equal bit for bit after four
    ticks, pins, floor contacts and two materials included."""
    res = []
    for flag in ("0", "1"):
        tune("PIES_PD_ROW_DICT", flag)
        g = pies.Solver(pd_options(pies, 6))
        g.create_tet_box(12, 12, 14, translation=(0.0, 0.02, 0.0), w=1.0, volume=True, triangles=True)
        g.create_tet_box(6, 5, 7, translation=(20.0, 0.5, 1.0), w=3.0, volume=True, triangles=True)
        scenes.perturb(g, 17, 0.05)
        g.set_prev_positions(g.positions)
        g.add_position(np.arange(0, 35, dtype=np.uint32), 2.0)
        g.tick(4)
        assert not g.failed
        stencils = g.count(pies.ROW_STENCILS)
        assert (stencils > 0) == (flag == "1") and stencils * 8 <= g.count(pies.NODES), stencils
        res.append((g.positions, g.velocities, g.pcg_stats()))
        g.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # (the statistics are sums over the rows in the order the kernels' threads own them - the windowed matrix that serves a scene
    # without a dictionary since round 5 deals its rows differently -, so the largest residual agrees to rounding, not to the bit)
    assert res[0][2][1:] == res[1][2][1:] and abs(res[0][2][0] - res[1][2][0]) <= 1e-3 * res[1][2][0]


def test_pd_flattened_and_inverted_elements(pies, oracle):
    """Elements with a collapsed direction (a layer of nodes pressed into the layer below: s = 0, the rotation is completed from
    the other two directions, dev_math.h svd3_recompose) and inverted ones (a node pushed through the opposite face: the
    smallest singular value is negated, Constraints.cpp:76-128).  The packed local step hands an element with a collapsed
    direction to the scalar routine; its neighbour in the lane stays on the packed path."""
    def build(s):
        s.create_tet_box(4, 4, 4, translation=(0.0, 3.0, 0.0), w=1.0, volume=True, triangles=True)
        p = s.positions
        top = np.isclose(p[:, 1], 3.0 + 3.0)
        p[top, 1] = 3.0 + 2.0                 # the top layer lies in the one below: its elements have no volume
        inv = np.argmin(np.abs(p - np.array([1.0, 3.0, 1.0])).sum(1))
        p[inv] += np.array([0.0, 1.6, 0.0], np.float32)   # through the face above: inverted elements around it
        s.set_positions(p)
        s.set_prev_positions(p)
    g = pies.Solver(pd_options(pies, 6))
    o = oracle.OracleSolver(pd_options(oracle, 6))
    g.set_pcg(3e-7, 64)
    for s in (g, o):
        build(s)
    TOL = tol_for(o.positions)                # (measured: 1.4e-5 of a lattice spacing, a fifth of the tolerance)
    for t in range(3):
        g.tick(); o.tick()
        assert np.isfinite(g.positions).all() and not g.failed
        within("pd_flattened_inverted", g, o, TOL)


def test_config3_l100k_against_oracle(pies, oracle):
    """BASELINE config 3 at full size (20x20x250 beam, PD, strain + volume constraints, 10 local/global iterations, end
    cap pinned, floor + point-triangle pipeline on): two ticks against the oracle's direct fp32 solve (banded Cholesky,
    bandwidth 421 after sorting along the beam), same tolerance as the small cases."""
    g = pies.Solver(pd_options(pies, 10))
    o = oracle.OracleSolver(pd_options(oracle, 10))
    o64 = oracle.OracleSolver(pd_options(oracle, 10))  # the yardstick: the same loop with the global solve in double
    o64.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    for s in (g, o, o64):
        build_pd_beam(s, scenes.L100K, translation=(0.0, 2.0, 0.0))
        scenes.perturb(s, 21, 0.03)
        s.set_prev_positions(s.positions)
    assert g.count(pies.TET) == g.count(pies.VOLUME) == 539334
    for t in range(2):
        g.tick(); o.tick(); o64.tick()
        yardstick("config3_l100k", g, o, o64, names=("positions", "prev_positions", "velocities"))
    res, iters_used, solves = g.pcg_stats()
    assert solves == 10 and res <= 3e-7 * 1.0001
    assert g.pcg_health()["short_solves"] == 0 and not g.failed


def test_config3_l100k_pd_properties(pies):
    """BASELINE config 3 at full size: size-independent properties: CG reaches the requested residual, state stays
    finite, clamped cap stays put, the free part sags."""
    g = pies.Solver(pd_options(pies, 10))
    build_pd_beam(g, scenes.L100K, translation=(0.0, 2.0, 0.0))
    p0 = g.positions
    g.tick(3)
    p = g.positions
    assert np.isfinite(p).all()
    res, iters_used, solves = g.pcg_stats()
    assert solves == 10 and res <= 1e-6 and iters_used < 12
    W, H, D = scenes.L100K
    cap = np.array([0 + D * (j + H * i) for i in range(W) for j in range(H)])
    assert np.abs(p[cap] - p0[cap]).max() < 0.02          # pinned end cap barely moves
    assert p[:, 1].mean() < p0[:, 1].mean()               # the free part sags under gravity


def _region(center, half):
    m = np.zeros((4, 4), np.float32)  # column-major: m[col][row]
    m[0, 0], m[1, 1], m[2, 2], m[3, 3] = half[0], half[1], half[2], 1.0
    m[3, :3] = center
    return m.reshape(16)


def test_pd_bend_sheet(pies, oracle):
    g = pies.Solver(pd_options(pies, 5))
    o = oracle.OracleSolver(pd_options(oracle, 5))
    for s in (g, o):
        s.create_bend_sheet(7, 6, translation=(0, 3, 0), scale=1.0, w=0.8)
        scenes.perturb(s, 6, 0.05)
        s.set_prev_positions(s.positions)
        s.tick(4)
    assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions)


def test_pd_shape_matching_box_and_sheet(pies, oracle):
    """C6: createShapeMatchingBox (one constraint over 4x3x5 nodes) plus a second, overlapping cluster.
    (createShapeMatchingSheet builds planar patches whose Q = sum r r^T is singular, so the reference's own
    Qinv is non-finite there; the sheet is therefore only compared structurally, in test_host_logic.)"""
    g = pies.Solver(pd_options(pies, 5))
    o = oracle.OracleSolver(pd_options(oracle, 5))
    for s in (g, o):
        s.create_shape_matching_box((0, 1.0, 0), 4, 3, 5, 30.0)
        s.create_shape_matching_box((4, 0.4, 0), 6, 6, 6, 5.0)
        s.add_shape(np.arange(40, 100, dtype=np.uint32), 3.0)  # overlaps both boxes' node ranges
        s.add_triangles([[0, 1, 5], [60, 61, 70], [100, 130, 170]])  # lets the bodies feel the floor
        scenes.perturb(s, 8, 0.05)
        s.set_prev_positions(s.positions)
    assert g.count(pies.SHAPE) == o.count(oracle.SHAPE) == 3
    for k in range(g.count(pies.SHAPE)):
        assert np.array_equal(g.group_ids(pies.SHAPE, k), o.group_ids(oracle.SHAPE, k))
    for t in range(5):
        g.tick(); o.tick()
        assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions), t
    assert np.abs(g.velocities - o.velocities).max() <= tol_for(o.positions) / 0.012


def test_pd_fixed_and_linked_regions(pies, oracle):
    """C7 + region API: pin one end of a beam with a fixed region, move it, link a block with shape matching."""
    g = pies.Solver(pd_options(pies, 6))
    o = oracle.OracleSolver(pd_options(oracle, 6))
    fixed = _region((1.5, 2.5, 0.0), (2.0, 2.0, 0.6))
    linked = _region((1.5, 2.5, 6.0), (2.5, 2.5, 1.2))
    for s in (g, o):
        s.create_tet_box(4, 4, 9, translation=(0, 1.0, 0), w=1.0, volume=True, triangles=True)
        s.add_fixed_regions(fixed, 50.0)
        s.add_linked_regions(linked, 10.0)
    assert g.count(pies.GOAL) == 1 and g.count(pies.SHAPE) == 1
    assert np.array_equal(g.group_ids(pies.GOAL, 0), o.group_ids(oracle.GOAL, 0)) and len(g.group_ids(pies.GOAL, 0)) == 16
    assert np.array_equal(g.group_ids(pies.SHAPE, 0), o.group_ids(oracle.SHAPE, 0))
    for s in (g, o):
        s.tick(2)
    moved = _region((1.5 + 0.3, 2.5 + 0.2, 0.1), (2.0, 2.0, 0.6))  # drag the clamped end
    for s in (g, o):
        s.update_fixed_regions(moved)
        s.tick(3)
    assert np.abs(g.positions - o.positions).max() <= tol_for(o.positions)
    cap = g.group_ids(pies.GOAL, 0)
    # w = 50 against m/h^2 ~ 7e3: the goal nodes are pulled towards the moved region, a little per tick
    assert 0.005 < g.positions[cap, 0].mean() - 1.5 < 0.3


def test_pd_unstructured_delaunay_beam(pies, oracle):
    """BASELINE configs 2-3 call for a tetgen beam (Src/PrimitiveUtilities.cpp:243-328 is the ingestion path): PD on an
    UNSTRUCTURED mesh - a Delaunay beam, one strain + one volume constraint per tetrahedron, the surface faces as collision
    triangles, the end cap pinned - takes the paths a lattice does not: per-element rest constants (no rest dictionary), the SELL
    matrix (no row dictionary), tiles cut by a Morton curve through irregular elements.  Five ticks against the oracle."""
    mesh = scenes.delaunay_beam((7, 6, 40))
    g = pies.Solver(pd_options(pies, 10))
    o = oracle.OracleSolver(pd_options(oracle, 10))
    for s in (g, o):
        scenes.build_unstructured_pd(s, mesh)
        scenes.perturb(s, 21, 0.02)
        s.set_prev_positions(s.positions)
    g.finalize()
    assert g.count(pies.REST_SETS) == 0 and g.count(pies.ROW_STENCILS) == 0 and g.count(pies.PD_TILES) > 0
    TOL = tol_for(o.positions)
    for t in range(5):
        g.tick(); o.tick()
        for name in ("positions", "velocities", "prev_positions"):
            within("pd_unstructured_7x6x40", g, o, TOL * (1.0 if name != "velocities" else 1.0 / 0.012), name)
    res, iters_used, solves = g.pcg_stats()
    assert res <= 1e-6 and solves == 10 and g.pcg_health()["short_solves"] == 0 and not g.failed


def test_pd_unstructured_full_size_properties(pies, tune):
    """The same kind of mesh at BASELINE's size (100k particles, ~590k element pairs), where the oracle's direct solve is out of
    reach of a test: size-independent properties instead.  (i) The tile-resident local step + the right-hand side inside the
    residual kernel against the per-(element, node) records + k_pd_rhs + the two-launch CG of round 3: the same arithmetic per
    element, other summation orders and another CG recurrence - agreement at rounding level; (ii) every solve below the
    tolerance; (iii) two runs of the same build are bit-identical."""
    mesh = scenes.delaunay_beam(scenes.L100K)

    def run(ticks=3):
        g = pies.Solver(pd_options(pies, 10))
        scenes.build_unstructured_pd(g, mesh)
        scenes.perturb(g, 21, 0.02)
        g.set_prev_positions(g.positions)
        g.tick(ticks)
        res, iters_used, solves = g.pcg_stats()
        out = (g.positions, g.velocities, res, g.pcg_health()["short_solves"], g.failed, g.count(pies.PD_TILES))
        g.close()
        return out
    a = run()
    b = run()
    assert a[5] > 0 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # deterministic
    assert a[2] <= 1e-6 and a[3] == 0 and not a[4] and np.isfinite(a[0]).all()
    tune("PIES_PD_TILE_ELEMS", "0")
    tune("PIES_PD_CG_SINGLE", "0")
    c = run()
    assert c[5] == 0 and c[2] <= 1e-6 and c[3] == 0
    d = float(np.abs(a[0] - c[0]).max())
    record("pd_unstructured_l100k_tiles_vs_records", "positions", d, tol_for(c[0]))
    assert d <= tol_for(c[0]), d


@pytest.mark.gpu
def test_pd_node_pair_collision_constraints(pies, oracle):
    """K2, the PD node-node CollisionConstraint (Src/CollisionConstraint.cpp:7-65; friction loop Solver.cpp:398-428) - an EXTENSION
    container here, because the reference never creates one (its only source, _parallelComputeCollisions, is never called, and
    tickPD calls none of the type's methods).  Scene: two small tet boxes pushed into each other with the overlapping node pairs
    listed by hand, plus loose spheres that overlap, touch and miss.  Device against the oracle's restatement, ticks of its own."""
    def build(s):
        s.create_tet_box(3, 3, 3, translation=(0.0, 1.0, 0.0), w=1.0)
        s.create_tet_box(3, 3, 3, translation=(2.6, 1.1, 0.2), w=1.0)           # overlaps the first box's x = 2 face
        s.addNodes(np.float32([[6.0, 2.0, 0.0], [6.6, 2.1, 0.1], [8.0, 2.0, 0.0], [9.0, 2.0, 0.0], [12.0, 2.0, 0.0], [14.5, 2.0, 0.0]]))
        r = s.radii
        r[:] = 0.5
        s.set_radii(r)
        v = s.velocities
        v[27:54, 0] = -1.5   # the second box moves into the first
        v[54:, 1] = 0.7
        s.set_velocities(v)
        s.set_prev_positions(s.positions)
        p = s.positions
        pairs = [(i, 27 + j) for i in range(27) for j in range(27) if np.linalg.norm(p[i] - p[27 + j]) < 1.2]
        pairs += [(54, 55), (56, 57), (58, 59)]  # overlapping, exactly touching, apart
        s.add_node_pairs(np.uint32(pairs))
        return len(pairs)
    g = pies.Solver(pd_options(pies, 6, friction=0.3, staticFrictionThreshold=0.05))
    o = oracle.OracleSolver(pd_options(oracle, 6, friction=0.3, staticFrictionThreshold=0.05))
    o64 = oracle.OracleSolver(pd_options(oracle, 6, friction=0.3, staticFrictionThreshold=0.05))
    o64.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    n = [build(s) for s in (g, o, o64)]
    assert n[0] == n[1] > 20 and g.count(pies.NODE_PAIRS) == o.count(oracle.NODE_PAIRS) == n[0]
    assert g.ids(pies.NODE_PAIRS).shape == (n[0], 2)
    g.set_pcg(3e-7, 256)  # w = 1e5 on the diagonal beside elastic terms of order 1: more CG iterations than the default cap
    start = g.positions.copy()
    for t in range(4):
        g.tick(); o.tick(); o64.tick()
        yardstick("pd_node_pairs", g, o, o64, names=("positions", "velocities"))
    # the constraint did something: the overlapping loose pair was pushed apart to about the sum of the radii, the distant pair not
    p = g.positions
    assert np.linalg.norm(p[54] - p[55]) > np.linalg.norm(start[54] - start[55]) + 0.2
    assert abs(np.linalg.norm(p[58] - p[59]) - np.linalg.norm(start[58] - start[59])) < 1e-3
    assert g.pcg_health()["short_solves"] == 0 and not g.failed


@pytest.mark.gpu
@pytest.mark.parametrize("sort,kernels,chunk,halo32", [("0", "3", "256", ""), ("1", "3", "256", ""), ("1", "1", "64", ""), ("0", "7", "1024", ""),
                                                     ("1", "7", "64", ""), ("1", "7", "256", "1")])
def test_windowed_matrix_changes_nothing(pies, tune, sort, kernels, chunk, halo32):
    """The windowed SELL matrix (its only coverage was the row-dictionary test) against the plain SELL arrays
    (PIES_PD_WINDOW=0) on an unstructured beam: rows sorted by length or not (PIES_PD_WINDOW_SORT), the window taken by the
    iterations only / + the first product / + the residual kernel with its fp64 sums and the right-hand side parked in LDS
    (PIES_PD_WINDOW_KERNELS 1 / 3 / 7), chunks of 64 / 256 / 1 024 rows (a partial last chunk, and at 1 024 rows a window above 64 KB
    of dynamic LDS on this mesh), the halo list in 16-bit offsets or 32-bit indices.  Entries keep their order inside a row, so a row's sum is the same fused multiply-add chain in
    every form: positions and velocities equal bit for bit after three ticks."""
    mesh = scenes.delaunay_beam((6, 5, 30), seed=11)
    res = []
    for window in ("0", "1"):
        tune("PIES_PD_WINDOW", window)
        tune("PIES_PD_WINDOW_SORT", sort)
        tune("PIES_PD_WINDOW_KERNELS", kernels)
        tune("PIES_CG_CHUNK_ROWS", chunk)
        tune("PIES_PD_WINDOW_HALO32", halo32)  # (the halo list with 32-bit entries: what a mesh whose chunks reach further than 65 535 rows takes)
        g = pies.Solver(pd_options(pies, 6))
        scenes.build_unstructured_pd(g, mesh)
        scenes.perturb(g, 5, 0.03)
        g.set_prev_positions(g.positions)
        g.finalize()
        assert (g.count(pies.PD_WINDOW_ENTRIES) > 0) == (window == "1") and g.count(pies.ROW_STENCILS) == 0
        g.tick(3)
        assert not g.failed and g.pcg_health()["short_solves"] == 0
        res.append((g.positions, g.velocities))
        g.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), float(np.abs(res[0][0] - res[1][0]).max())
