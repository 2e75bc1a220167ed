"""Pins the CPU oracle (oracle/pies_oracle.cpp) -- the reference has no tests or fixtures ("parity
unpinned"), so the pins are fp64 golden vectors (tests/golden/make_golden.py) and analytic known answers.

Tolerances: a single projection against fp64: 2e-5 absolute on O(1) values (fp32 SVD); whole-loop tiny
scenes: 2e-4 * spacing after <= 6 ticks (fp32 vs fp64 drift)."""
import os

import numpy as np
import pytest

import oracle_api as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_svd_fixed_against_fp64():
    d = load("svd_fixed.npz")
    worst = 0.0
    for A, exp in zip(d["A"], d["expected"]):
        # the tet functor on a unit reference tet at the origin reproduces fixed(F): x = (0, F cols), Qinv = I
        x = np.vstack([np.zeros(3, np.float32), A.T])
        out = O.project_tet(x, np.eye(3, dtype=np.float32).reshape(9), float(d["lo"]), float(d["hi"]))
        Fh = out[1:].T
        cond = np.linalg.cond(A.astype(np.float64))
        tol = 2e-5 if cond < 1e3 else 2e-4
        err = np.abs(Fh - exp).max()
        assert err <= tol, (err, cond)
        worst = max(worst, err)
    assert worst > 0  # not trivially identical


@pytest.mark.parametrize("name,fn", [("tet_projection.npz", O.project_tet), ("volume_projection.npz", O.project_volume)])
def test_tet_and_volume_projection_against_fp64(name, fn):
    d = load(name)
    for x, q, exp in zip(d["x"], d["qinv"], d["expected"]):
        out = fn(x, q, float(d["lo"]), float(d["hi"]))
        scale = max(1.0, np.abs(exp).max())
        assert np.abs(out - exp).max() <= 5e-5 * scale


def test_distance_projection_against_fp64():
    d = load("distance_projection.npz")
    for x, t, exp in zip(d["x"], d["target"], d["expected"]):
        out = O.project_distance(x, float(t))
        assert np.abs(out - exp).max() <= 1e-5
        if np.linalg.norm(x[1] - x[0]) > 1e-3:
            assert abs(np.linalg.norm(out[1] - out[0]) - t) <= 1e-5  # |pa - pb| = target
        assert np.array_equal(out[1], x[1])  # node b never moves (Constraints.cpp:34-36)


def test_bend_projection_against_fp64():
    d = load("bend_projection.npz")
    for x, im, a, exp in zip(d["x"], d["invMass"], d["angle"], d["expected"]):
        out = O.project_bend(x, im, float(a))
        scale = max(1.0, np.abs(exp - x).max())
        assert np.abs(out - exp).max() <= 2e-4 * scale


# ---- analytic known answers --------------------------------------------------------------------
REST = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])


def test_rest_tet_projects_to_identity_columns():
    q, AtA = O.tet_rest(REST)
    assert np.allclose(q.reshape(3, 3), np.eye(3))
    out = O.project_tet(REST, q)
    assert np.allclose(out, np.vstack([np.zeros(3), np.eye(3)]), atol=1e-6)
    # A = [0; Qinv^T D] with D = [-1 1 0 0; -1 0 1 0; -1 0 0 1]  =>  A^T A for the unit tet
    A = np.zeros((4, 4)); A[1:] = np.array([[-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]])
    assert np.allclose(AtA, A.T @ A)


@pytest.mark.parametrize("s,expect", [(0.5, 0.8), (0.9, 0.9), (1.7, 1.0)])
def test_uniformly_scaled_tet_is_clamped(s, expect):
    q, _ = O.tet_rest(REST)
    out = O.project_tet(REST * np.float32(s), q)
    assert np.allclose(out[1:], expect * np.eye(3), atol=2e-6)


def test_rotated_tet_keeps_rotation():
    q, _ = O.tet_rest(REST)
    c, s = np.cos(0.7), np.sin(0.7)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    x = (REST.astype(np.float64) @ R.T * 1.3 + [3, 1, -2]).astype(np.float32)
    out = O.project_tet(x, q)
    assert np.allclose(out[1:].T, R, atol=3e-6)  # F = 1.3 R -> clamp to 1.0 -> R (columns)


def test_mirrored_tet_flips_smallest_singular_value():
    q, _ = O.tet_rest(REST)
    # F = diag(1,1,-0.9) = U S V^T with U = diag(1,1,-1): negating sigma_3 (Constraints.cpp:106-108)
    # un-inverts the element: Fhat = U diag(1,1,-0.9) V^T = diag(1,1,+0.9), det > 0
    x = REST.copy(); x[3, 2] = -0.9
    out = O.project_tet(x, q)
    assert np.allclose(out[1:].T, np.diag([1, 1, 0.9]), atol=2e-6)


def test_volume_projection_restores_unit_volume():
    q, _ = O.tet_rest(REST)
    x = REST * np.float32([1.2, 0.9, 1.1])
    out = O.project_volume(x, q, 1.0, 1.0)
    assert abs(np.linalg.det(out[1:].T.astype(np.float64)) - 1.0) < 1e-5
    assert np.allclose(O.project_volume(REST, q, 1.0, 1.0)[1:], np.eye(3), atol=1e-6)  # idempotent at rest


def test_node_range_matches_definition():
    # Solver.cpp:877-901 with r = 0.5, pad 0.5, scale 2: R = 0.5 grid units
    r = O.node_range([2.2, -0.1, 7.9], 0.5, 2.0)
    assert r.tolist() == [0, -1, 3, 2, 2, 2]  # e.g. y: min = -0.05 - 0.5 = -0.55 -> cell -1, ceil(0.45 + 1) = 2
    assert O.node_range([0, 0, 0], 200.0, 2.0)[3:].tolist() == [0, 0, 0]  # > 50 cells: empty range


# ---- whole-loop restatements on tiny scenes -----------------------------------------------------
def _tiny_pbd(d):
    W, H, D = (int(v) for v in d["dims"])
    o = O.OracleSolver(solver=O.PBD, iterations=int(d["iterations"]))
    o.create_tet_box(W, H, D, translation=d["translation"], scale=float(d["spacing"]), w=float(d["w_tet"]), volume=False,
                     triangles=False)
    o.create_box(W, H, D, scale=float(d["spacing"]), w=float(d["w_dist"]), existing_offset=0, triangles=False)
    o.set_radii(np.full(W * H * D, d["radius"], np.float32))
    o.set_flag(O.FLAG_NODE_COLLISIONS, int(d["collisions"]))
    return o


@pytest.mark.parametrize("coll", [0, 1])
def test_pbd_tiny_scene_against_fp64_restatement(coll):
    """One tick at a time from the stored float32 state (the PBD dynamics amplify rounding noise ~100x per
    tick, see make_golden.py); tolerance 2e-4 * spacing per tick."""
    d = load("pbd_tiny_coll%d.npz" % coll)
    o = _tiny_pbd(d)
    for t in range(len(d["pos"]) - 1):
        o.set_positions(d["pos"][t]); o.set_velocities(d["vel"][t])
        o.tick()
        assert np.abs(o.positions - d["pos"][t + 1]).max() <= 2e-4, t
        assert np.abs(o.velocities - d["vel"][t + 1]).max() <= 2e-4 / 0.012, t
    if coll:
        assert o.collision_pairs > 0


def test_pd_tiny_scene_against_fp64_restatement():
    d = load("pd_tiny.npz")
    W, H, D = (int(v) for v in d["dims"])
    o = O.OracleSolver(solver=O.PD, iterations=int(d["iterations"]))
    o.create_tet_box(W, H, D, translation=d["translation"], w=1.0, volume=True, triangles=True)
    o.add_position(d["pins"], float(d["w_pin"]))
    o.set_positions(d["pos"]); o.set_prev_positions(d["pos"]); o.set_velocities(d["vel"])
    for exp in d["expected"]:
        o.tick()
        assert np.abs(o.positions - exp).max() <= 2e-4
    assert o.count(O.STATICS) > 0  # the floor contacts were exercised


def test_pd_rest_state_is_a_fixed_point_without_gravity():
    o = O.OracleSolver(solver=O.PD, iterations=3, gravity=0.0)
    o.create_tet_box(3, 3, 3, translation=(0, 5, 0), w=1.0)
    p0 = o.positions
    o.tick(3)
    assert np.abs(o.positions - p0).max() < 5e-5  # fp32 round-off of the M/h^2-weighted solve, no drift


def test_pbd_free_fall_and_floor():
    o = O.OracleSolver(solver=O.PBD, iterations=2)
    o.addNodes([[0, 3, 0]])
    o.set_flag(O.FLAG_NODE_COLLISIONS, 0)
    o.tick()
    dt = 0.012
    assert np.allclose(o.positions[0], [0, 3 - 10 * dt * dt, 0], atol=1e-6)
    assert np.allclose(o.velocities[0], [0, (1 - 0.006) * (-10 * dt), 0], atol=1e-5)
    o.tick(400)
    assert abs(o.positions[0, 1] - 0.5) < 1e-6  # rests on the floor at y = radius


def test_two_overlapping_spheres_are_pushed_apart():
    # Solver.cpp:85-130: each ordered pair in each shared cell relaxes 85 % of the overlap
    o = O.OracleSolver(solver=O.PBD, iterations=1, gravity=0.0)
    o.addNodes([[0.0, 5, 0], [0.8, 5, 0]])
    o.tick()
    p = o.positions
    assert p[1, 0] - p[0, 0] > 0.8 and abs((p[0, 0] + p[1, 0]) - 0.8) < 1e-6
    assert o.collision_pairs >= 2


# ---- point-triangle CCD (D1) ----------------------------------------------------------------------
def test_point_triangle_ccd_against_fp64():
    d = load("point_triangle_ccd.npz")
    hits = 0
    for a, (hit, t) in zip(d["args"], d["expected"]):
        got, tt = O.point_triangle_ccd(*a, float(d["threshold"]))
        assert got == bool(hit), (a, hit, t, tt)
        if hit:
            hits += 1
            assert abs(tt - t) <= 2e-4  # bisection to float resolution vs numpy.roots
    assert 100 < hits < 300


def test_ccd_known_answers():
    b, c, dd = np.float32([0, 0, 0]), np.float32([1, 0, 0]), np.float32([0, 0, -1])  # normal (0, 1, 0)
    tri = lambda a0, a1: O.point_triangle_ccd(a0 - b, c - b, dd - b, a1 - b, c - b, dd - b, 0.1)
    hit, t = tri(np.float32([0.2, 1.0, -0.2]), np.float32([0.2, -1.0, -0.2]))  # straight through at t = 0.5
    assert hit and abs(t - 0.5) < 1e-6
    hit, t = tri(np.float32([0.2, 0.05, -0.2]), np.float32([0.2, 0.04, -0.2]))  # resting inside the threshold
    assert hit and t == 0.0
    assert not tri(np.float32([0.2, 0.5, -0.2]), np.float32([0.2, 0.3, -0.2]))[0]   # approaches, too far
    assert not tri(np.float32([2.0, 1.0, -0.2]), np.float32([2.0, -1.0, -0.2]))[0]  # crosses the plane outside
    assert not tri(np.float32([0.2, -0.05, -0.2]), np.float32([0.2, -0.04, -0.2]))[0]  # behind the triangle


def _two_boxes(o):
    o.create_tet_box(3, 3, 3, translation=(0, 0.02, 0), w=1.0)        # sits on the floor
    o.create_tet_box(3, 3, 3, translation=(0.4, 2.06, 0.3), w=1.0)   # just above it, offset, falling
    v = o.velocities
    v[27:, 1] = -2.0
    o.set_velocities(v)
    o.set_prev_positions(o.positions)


def test_pd_two_boxes_collide_through_triangle_ccd():
    """K3/H3/D1 in the oracle: a box dropped on another is caught by point-triangle contacts."""
    res = {}
    for tri in (1, 0):
        o = O.OracleSolver(solver=O.PD, iterations=6)
        _two_boxes(o)
        o.set_flag(O.FLAG_TRIANGLE_COLLISIONS, tri)
        contacts = 0
        for _ in range(12):
            o.tick()
            contacts += o.count(O.TRI_CONTACTS)
        res[tri] = (o.positions, contacts)
        assert not o.failed and np.isfinite(o.positions).all()
    assert res[0][1] == 0 and res[1][1] > 0
    # node 36 (bottom face of the upper box, over the lower box): held up by its contacts, free-falls without
    # (w = 1 against m/h^2 ~ 7e3 makes the boxes very soft, so only the contacting nodes are compared)
    assert res[1][0][36, 1] > res[0][0][36, 1] + 0.15
    assert res[1][0][36, 1] > res[1][0][:27, 1].max() - 0.1  # it rides on the lower box's top face


def test_multithreaded_batch_sweep_equals_the_sequential_sweep():
    """The all-cores CPU baseline of bench.py: conflict-free batches of a coloured device plan swept with OpenMP threads
    give the bits of the single-threaded sweep in the same order."""
    import scenes
    from pies_amd import capi
    g = capi.Solver(scenes.pbd_options(capi, 3), device=capi.DEVICE_NONE)
    scenes.build_beam(g, (5, 6, 9))
    g.set_schedule(capi.SCHEDULE_COLOURED)
    g.finalize()
    res = []
    for threads in (1, 4):
        o = O.OracleSolver(scenes.pbd_options(O, 3))
        scenes.build_beam(o, (5, 6, 9))
        scenes.perturb(o, 8, 0.05)
        o.set_flag(1, 0)
        for t in (capi.DISTANCE, capi.TET):
            o.permute(t, g.order(t))
            o.set_batches(t, g.batches(t))
        o.set_threads(threads)
        o.tick(3)
        res.append((o.positions.copy(), o.velocities.copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_svd_against_an_independent_fp32_svd():
    """The reference calls Eigen::JacobiSVD<Matrix3f> (absent here); the oracle and the device run their own one-sided Jacobi.
    Only U f(S) V^T is consumed, which does not depend on the algorithm - so ANOTHER fp32 SVD (LAPACK's sgesdd through numpy)
    must give the same projection up to fp32 rounding.  This measures that rounding-level gap between two fp32 SVD
    implementations, i.e. the size of the gap to expect against Eigen's."""
    d = load("svd_fixed.npz")
    lo, hi = np.float32(d["lo"]), np.float32(d["hi"])
    worst = 0.0
    for A in d["A"]:
        if np.linalg.cond(A.astype(np.float64)) >= 1e3:
            continue
        x = np.vstack([np.zeros(3, np.float32), A.T])
        ours = O.project_tet(x, np.eye(3, dtype=np.float32).reshape(9), float(lo), float(hi))[1:].T
        U, S, Vt = np.linalg.svd(A.astype(np.float32))
        Sn = np.clip(S, lo, hi)
        if np.linalg.det(A.astype(np.float64)) < 0:
            Sn[2] = -Sn[2]  # the smallest singular value is flipped (Constraints.cpp:106-108)
        theirs = (U * Sn) @ Vt
        worst = max(worst, float(np.abs(ours - theirs.astype(np.float32)).max()))
    print("oracle projection vs LAPACK fp32 SVD: max abs difference %.3g" % worst)
    assert 0 < worst <= 1e-5


def test_collision_orders_agree_where_order_cannot_matter():
    """The oracle's three node-node orders (0 the reference's loop, 1 the group order, 2 the pair order) execute the same
    visits in different orders.  When no node takes part in more than one overlapping pair the order can only matter through
    rounding (a node's visits to itself cancel up to an ulp, and the orders interleave them differently with the pair's visits):
    isolated pairs of overlapping spheres (and singles) far apart must come out the same to 1e-5 in all three, with the same
    number of resolved visits to within 2 % - the self visits of quirk Q3 and the repeated visits per shared cell included.  (Rule 0 looks the visiting node's range up from its LIVE position, SpatialHash.h:101-106, rules 1 and 2 from the
    position it was inserted with: a node that has crossed a cell boundary inside the iteration meets a partner a different
    number of times, so rule 0's count may differ by a fraction of a percent.)"""
    rng = np.random.default_rng(11)
    centres = np.stack(np.meshgrid(np.arange(6), np.arange(4), np.arange(5), indexing="ij"), -1).reshape(-1, 3) * 7.0 + [0.3, 3.0, 0.1]
    nodes = []
    for k, c in enumerate(centres):
        nodes.append(c + rng.uniform(-0.8, 0.8, 3))
        if k % 3 != 2:  # a partner 0.5 - 0.95 away (radius 0.5 each: overlapping), in a random direction
            d = rng.normal(size=3)
            nodes.append(nodes[-1] + d / np.linalg.norm(d) * rng.uniform(0.5, 0.95))
    p = np.float32(nodes)
    v = rng.uniform(-1, 1, p.shape).astype(np.float32)
    out = []
    for rule in (0, 1, 2):
        o = O.OracleSolver(solver=O.PBD, iterations=3, friction=0.2, staticFrictionThreshold=0.3)
        o.addNodes(p)
        o.set_velocities(v)
        o.set_flag(O.FLAG_COLLISION_RULE, rule)
        o.tick(2)
        out.append((o.positions, o.velocities, o.collision_pairs))
    assert out[0][2] > 500
    for rule in (1, 2):
        assert np.abs(out[0][0] - out[rule][0]).max() < 1e-5 and np.abs(out[0][1] - out[rule][1]).max() < 1e-3, rule
    # (a pair's overlap shrinks by 0.15 per resolved visit: the last of its visits see overlaps at rounding level and count or not)
    assert max(o[2] for o in out) - min(o[2] for o in out) < 0.02 * out[0][2], [o[2] for o in out]


def test_svd_closed_form_start_on_every_matrix_class():
    """Round 6's decomposition (ora_math.h svd3: closed-form eigenvector frame of A^T A, one rotation, three small-angle polishes,
    certifying sweeps) on the classes that could break a closed form: generic, near rest, double and triple singular values,
    nearly flat (sigma_3 down to 1e-5 sigma_1), needles, exactly rank-deficient (a zero column; a generic rank-2 matrix), zero,
    tiny and huge scale (outside the closed form's range: the plain iteration takes them).  For every matrix: A V = B to rounding,
    V orthonormal, the columns of B orthogonal to the certified tolerance, the singular values LAPACK's in fp64 - and the plain
    iteration from V = I (FLAG_SVD_PLAIN's routine through project_tet) gives the same projection."""
    rng = np.random.default_rng(77)

    def rot(n):
        q, _ = np.linalg.qr(rng.normal(size=(n, 3, 3)))
        return q

    def diag(s, n):
        return rot(n) @ (np.asarray(s, dtype=np.float64)[None, :, None] * rot(n).transpose(0, 2, 1))
    n = 300
    classes = {
        "generic": rng.normal(size=(n, 3, 3)),
        "near rest": rot(n) @ (np.eye(3) + 0.05 * rng.normal(size=(n, 3, 3))),
        "exact rotation": rot(n),
        "double": diag([1.3, 1.3, 0.7], n), "double low": diag([1.3, 0.7, 0.7], n), "triple": 1.3 * rot(n),
        "flat 5e-2": diag([1.3, 0.9, 0.05], n), "flat 1e-5": diag([1.3, 0.9, 1e-5], n), "needle": diag([1.3, 1e-4, 2e-4], n),
        "zero column": rng.normal(size=(n, 3, 3)) * np.array([1.0, 0.0, 1.0])[None, None, :],
        "rank 2": diag([1.3, 0.9, 0.0], n), "zero": np.zeros((4, 3, 3)),
        "tiny": 1e-12 * rng.normal(size=(n, 3, 3)), "huge": 1e6 * rng.normal(size=(n, 3, 3)),
    }
    tol = 4.76837158203125e-07
    for name, As in classes.items():
        for A in As.astype(np.float32):
            s, b, v = O.svd3(A)  # b[i] = column i of A V, v[i] = column i of V
            scale = max(float(np.abs(A).max()), 1e-30)
            assert np.isfinite(s).all() and np.isfinite(b).all() and np.isfinite(v).all(), name
            assert np.abs(v @ v.T - np.eye(3)).max() < 2e-6, (name, "V orthonormal")
            assert np.abs(A.astype(np.float64) @ v.T.astype(np.float64) - b.T).max() < 4e-6 * scale, (name, "A V = B")
            for i, j in ((0, 1), (0, 2), (1, 2)):
                ni, nj = float(b[i] @ b[i]), float(b[j] @ b[j])
                if ni * nj > 1e-36:  # (numerically zero columns are exempt, like in the routine)
                    assert abs(float(b[i] @ b[j])) <= 1.05 * tol * np.sqrt(ni * nj) + 1e-18, (name, i, j)
            if name not in ("tiny",):  # (below 1e-18 a column counts as collapsed: s = 0 by contract)
                ref = np.linalg.svd(A.astype(np.float64), compute_uv=False)
                assert np.abs(np.sort(s)[::-1] - ref).max() < 3e-6 * max(ref[0], 1e-30) + 1e-18, (name, s, ref)
    # the plain iteration and the closed-form start give the same projection (up to fp32 rounding)
    x = np.vstack([np.zeros(3, np.float32), classes["flat 5e-2"][0].astype(np.float32).T])
    q = np.eye(3, dtype=np.float32).reshape(9)
    o = O.OracleSolver(O.Options())
    a = O.project_tet(x, q, 0.8, 1.0)
    o.set_flag(O.FLAG_SVD_PLAIN, 1)
    try:
        bplain = O.project_tet(x, q, 0.8, 1.0)
    finally:
        o.set_flag(O.FLAG_SVD_PLAIN, 0)
    assert np.abs(a - bplain).max() < 2e-6
