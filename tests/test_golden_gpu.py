"""The committed fp64 golden vectors (tests/golden/*.npz, generator tests/golden/make_golden.py) run through the HIP path.

tests/test_oracle_golden.py pins the CPU oracle with these vectors; the device's 3x3 SVD, its CCD and its projections share
their hand-made algorithms with that oracle, so device == oracle says nothing about the algorithms themselves.  Here every
record goes through the C ABI - one-constraint scenes built with the bulk raw ingestion (pies_add_*_constraints +
pies_set_rest), one tick - and is compared with the fp64 expectation at the tolerance the oracle's test uses:

* PBD (Src/Solver.cpp:58-75 through Include/Pies/Constraints.h:121-129): w = 1, one iteration, no gravity: the tick leaves
  pos + 1 * (projection - pos);  tetrahedral strain (Src/Constraints.cpp:76-128), distance (:11-37), bend (:312-366), in all
  three schedules (k_wave, k_tet / k_distance / k_bend, k_layer);
* PD local steps (Src/Solver.cpp:270-349, Src/Constraints.cpp:205-255): one local/global iteration of a scene of independent
  elements is a 4 x 4 (2 x 2) solve per element whose right-hand side holds the projection: positions against the fp64 solve
  with the golden projections, for the strain and the volume step alone (k_pd_local_tet), the fused pair (tiles / packed) and
  distance / bend;
* point-triangle CCD (Src/CollisionDetection.cpp:227-302): two-triangle scenes, the contact list (pies_get_tri_contacts)
  against the fp64 decisions;
* the whole-loop PBD restatement (pbd_tiny_coll{0,1}.npz) tick by tick under schedule EXACT, pd_tiny.npz under PD.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H = np.float32(0.012)  # fixedTimestepSize, one substep


def load(name):
    return np.load(os.path.join(G, name))


def golden_module():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SCHEDULES = ["exact", "coloured", "layered"]


def pbd_solver(pies, schedule):
    # no gravity, the floor far below, no node-node pass: one iteration of one container is the whole tick
    g = pies.Solver(pies.Options(solver=pies.PBD, iterations=1, timeSubsteps=1, fixedTimestepSize=float(H), gravity=0.0,
                                 floorHeight=-1.0e6, damping=0.0), device=0)
    g.set_flag(pies.FLAG_NODE_COLLISIONS, 0)
    g.set_schedule({"exact": pies.SCHEDULE_EXACT, "coloured": pies.SCHEDULE_COLOURED, "layered": pies.SCHEDULE_LAYERED}[schedule])
    return g


def blend(x, proj):
    """Constraints.h:125-128 with w = 1, in fp32 like the device: pos += 1 * (proj - pos) moves pos to proj up to an ulp of |pos|"""
    return np.abs(x).max() * 2.4e-7


# ---- PBD projections -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("schedule", SCHEDULES)
def test_tet_strain_projection_goldens_through_pbd(pies, schedule):
    d = load("tet_projection.npz")
    x, q, exp = d["x"], d["qinv"], d["expected"]
    n = len(x)
    g = pbd_solver(pies, schedule)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01)
    g.add_tet(np.arange(4 * n, dtype=np.uint32).reshape(n, 4), 1.0, float(d["lo"]), float(d["hi"]))
    g.set_rest(pies.TET, q)  # the golden's rest state (the factory took it from the deformed positions)
    assert np.array_equal(g.rest(pies.TET), q)
    g.tick(1)
    out = g.positions.reshape(n, 4, 3)
    g.close()
    worst = 0.0
    for k in range(n):
        scale = max(1.0, np.abs(exp[k]).max())
        err = np.abs(out[k] - exp[k]).max()
        assert err <= 5e-5 * scale + blend(x[k], exp[k]), (k, err)
        worst = max(worst, err)
    assert worst > 0


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_svd_goldens_through_pbd(pies, schedule):
    """svd_fixed.npz: fixed(F) of 600 matrices (well and ill conditioned, inverted, nearly equal singular values) - a unit
    reference tetrahedron at the origin (Qinv = I) whose edges are the columns of A reproduces it"""
    d = load("svd_fixed.npz")
    A, exp = d["A"], d["expected"]
    n = len(A)
    x = np.zeros((n, 4, 3), np.float32)
    x[:, 1:] = np.transpose(A, (0, 2, 1))
    g = pbd_solver(pies, schedule)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01)
    g.add_tet(np.arange(4 * n, dtype=np.uint32).reshape(n, 4), 1.0, float(d["lo"]), float(d["hi"]))
    g.set_rest(pies.TET, np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)))
    g.tick(1)
    out = g.positions.reshape(n, 4, 3)
    g.close()
    for k in range(n):
        Fh = out[k, 1:].T
        cond = np.linalg.cond(A[k].astype(np.float64))
        tol = 2e-5 if cond < 1e3 else 2e-4
        assert np.abs(Fh - exp[k]).max() <= tol + blend(x[k], exp[k]), (k, cond)
        assert np.abs(out[k, 0]).max() <= 1e-6  # the first node goes to the origin (Constraints.cpp:124)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_distance_projection_goldens_through_pbd(pies, schedule):
    d = load("distance_projection.npz")
    x, t, exp = d["x"], d["target"], d["expected"]
    n = len(x)
    g = pbd_solver(pies, schedule)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01)
    g.add_distance(np.arange(2 * n, dtype=np.uint32).reshape(n, 2), 1.0)
    g.set_rest(pies.DISTANCE, t)
    g.tick(1)
    out = g.positions.reshape(n, 2, 3)
    g.close()
    for k in range(n):
        assert np.abs(out[k] - exp[k]).max() <= 1e-5 + blend(x[k], exp[k]), k
        assert np.array_equal(out[k, 1], x[k, 1])  # node b never moves (Constraints.cpp:34-36)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_bend_projection_goldens_through_pbd(pies, schedule):
    d = load("bend_projection.npz")
    x, im, a, exp = d["x"], d["invMass"], d["angle"], d["expected"]
    n = len(x)
    g = pbd_solver(pies, schedule)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01, invMass=im.reshape(-1))
    g.add_bend(np.arange(4 * n, dtype=np.uint32).reshape(n, 4), 1.0)
    g.set_rest(pies.BEND, a)
    g.tick(1)
    out = g.positions.reshape(n, 4, 3)
    g.close()
    for k in range(n):
        scale = max(1.0, np.abs(exp[k] - x[k]).max())
        assert np.abs(out[k] - exp[k]).max() <= 2e-4 * scale + blend(x[k], exp[k]), k


# ---- PD local steps --------------------------------------------------------------------------------------------------------
def tet_A(q):
    """A = [0; Qinv^T D] (Constraints.cpp:157-176) from a column-major Qinv, fp64"""
    Qm = q.astype(np.float64).reshape(3, 3)  # Qm[r][k] = Qinv[col r][row k]
    Dm = np.array([[-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]], dtype=np.float64)
    A = np.zeros((4, 4))
    A[1:] = Qm @ Dm
    return A


def pd_solver(pies):
    g = pies.Solver(pies.Options(solver=pies.PD, iterations=1, timeSubsteps=1, fixedTimestepSize=float(H), gravity=0.0,
                                 floorHeight=-1.0e6, damping=0.0), device=0)
    g.set_flag(pies.FLAG_TRIANGLE_COLLISIONS, 0)
    g.set_pcg(1e-7, 128)
    return g


def inertia(inv_mass):
    """diag of Solver.cpp:179-182 as the device computes it: 1 / (invMass * h^2) in fp32"""
    return (np.float32(1.0) / (np.float32(inv_mass) * (H * H))).astype(np.float64)


# fp32 noise of an element's 4 x 4 solve with the inertia term at the size of the constraint term (condition number 2-3): the
# matrix entries, the right-hand side and the CG (relative residual 1e-7) each contribute a few ulp of the largest coordinate
SOLVE_NOISE = 1.5e-6


@pytest.mark.parametrize("which", ["strain", "volume", "pair"])
def test_tet_and_volume_projection_goldens_through_the_pd_local_step(pies, which):
    """One local/global iteration of independent elements: x_new = (m/h^2 + sum A^T A)^-1 (m/h^2 x + sum A^T p) with the projections p
    of the golden files (fp64).  Every element's nodes get the mass that puts m/h^2 at the largest eigenvalue of its sum A^T A:
    the 4 x 4 system then has condition number ~ 2, and an error dp of the device's projection moves x_new by K^-1 A^T dp - the
    test asks for that to stay below what the oracle's golden tolerance (5e-5 x scale on p) allows, plus the solve's fp32 noise."""
    ds, dv = load("tet_projection.npz"), load("volume_projection.npz")
    assert np.array_equal(ds["x"], dv["x"]) and np.array_equal(ds["qinv"], dv["qinv"])
    x, q = ds["x"], ds["qinv"]
    n = len(x)
    ids = np.arange(4 * n, dtype=np.uint32).reshape(n, 4)
    # (on, golden projections, tolerance on p: the oracle's test passes both at 5e-5 x scale.  The device's PD volume step computes
    # computeD's ten fixed-point iterations (Constraints.cpp:186-203) with one division and three products instead of three divisions
    # and shares one recomposition with the strain step: on the records the iteration has not converged on - a
    # uniformly compressed element, sigma = 0.3 -> 1 - the different rounding is carried through the ten iterations and shows as up to
    # 1.3e-4 against 5e-5 for the reference's own order of operations; PD parity is by tolerance: four times the oracle's)
    parts = [(which in ("strain", "pair"), ds["expected"], 5e-5), (which in ("volume", "pair"), dv["expected"], 2e-4)]
    AtA = np.array([sum(tet_A(q[k]).T @ tet_A(q[k]) for on, _, _ in parts if on) for k in range(n)])
    lam = np.array([np.linalg.eigvalsh(M)[-1] for M in AtA])
    inv_mass = (1.0 / (lam * float(H) ** 2)).astype(np.float32)
    g = pd_solver(pies)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01, invMass=np.repeat(inv_mass, 4))
    if parts[0][0]:
        g.add_tet(ids, 1.0, float(ds["lo"]), float(ds["hi"]))
        g.set_rest(pies.TET, q)
    if parts[1][0]:
        g.add_volume(ids, 1.0, float(dv["lo"]), float(dv["hi"]))
        g.set_rest(pies.VOLUME, q)
    g.tick(1)
    out = g.positions.reshape(n, 4, 3)
    if which == "pair":
        assert g.launch_counts().get("pd_local_volume", 0) == 0  # the fused strain + volume step ran (tiles or packed pairs)
    g.close()
    sharp = 0
    worst = []
    for k in range(n):
        A = tet_A(q[k])
        m = float(inertia(inv_mass[k]))
        K = m * np.eye(4) + AtA[k]
        rhs = m * x[k].astype(np.float64)
        ptol = 0.0
        for on, exp, tp in parts:
            if on:
                rhs = rhs + A.T @ exp[k]
                ptol += tp * max(1.0, np.abs(exp[k]).max())
        want = np.linalg.solve(K, rhs)
        gain = np.abs(np.linalg.solve(K, A.T)).sum(axis=1).max()  # |dx|_inf <= gain |dp|_inf (per projection)
        tol = gain * ptol + SOLVE_NOISE * max(np.abs(want).max(), np.abs(x[k]).max())
        err = np.abs(out[k] - want).max()
        worst.append((err / tol, k, err, tol))
        sharp += np.abs(want - x[k]).max() > 50 * tol  # the element moves by far more than the tolerance: the check has teeth
    worst.sort(reverse=True)
    print("PD local step [%s]: largest error / tolerance over %d records:" % (which, n), ["%.2f (record %d)" % (w[0], w[1]) for w in worst[:5]])
    assert worst[0][0] <= 1.0, worst[:5]
    assert sharp > 0.8 * n


def test_distance_projection_goldens_through_the_pd_local_step(pies):
    d = load("distance_projection.npz")
    x, t, exp = d["x"], d["target"], d["expected"]
    n = len(x)
    INV_MASS = np.float32(1.0) / (H * H)  # m / h^2 = 1 against A^T A = [[.5, -.5], [-.5, .5]]
    g = pd_solver(pies)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01, invMass=float(INV_MASS))
    g.add_distance(np.arange(2 * n, dtype=np.uint32).reshape(n, 2), 1.0)
    g.set_rest(pies.DISTANCE, t)
    g.tick(1)
    out = g.positions.reshape(n, 2, 3)
    g.close()
    m = float(inertia(INV_MASS))
    A = np.array([[0.5, -0.5], [-0.5, 0.5]])  # A = B (Constraints.cpp:44-52)
    for k in range(n):
        want = np.linalg.solve(m * np.eye(2) + A.T @ A, m * x[k].astype(np.float64) + A.T @ (A @ exp[k]))
        assert np.abs(out[k] - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), k


def test_bend_projection_goldens_through_the_pd_local_step(pies):
    d = load("bend_projection.npz")
    x, im, a, exp = d["x"], d["invMass"], d["angle"], d["expected"]
    n = len(x)
    w = 7000.0  # of the order of m / h^2 = 3 500 ... 14 000 for these inverse masses
    g = pd_solver(pies)
    g.add_nodes_raw(x.reshape(-1, 3), radius=0.01, invMass=im.reshape(-1))
    g.add_bend(np.arange(4 * n, dtype=np.uint32).reshape(n, 4), w)
    g.set_rest(pies.BEND, a)
    g.tick(1)
    out = g.positions.reshape(n, 4, 3)
    g.close()
    for k in range(n):
        m = inertia(im[k])[:, None]  # A = B = I (Constraints.cpp:376-384): one scalar equation per node
        want = (m * x[k].astype(np.float64) + w * exp[k]) / (m + w)
        scale = max(1.0, np.abs(exp[k] - x[k]).max())
        assert np.abs(out[k] - want).max() <= 2e-4 * scale, k


# ---- point-triangle CCD ----------------------------------------------------------------------------------------------------
def test_point_triangle_ccd_goldens_through_the_contact_list(pies, tune):
    """Every golden case (a point a against a moving triangle b, c, d, relative to b) as a scene of two triangles: (b, c, d) with b
    at the origin - the differences the device forms are the golden's bits - and (a, e, f) with e, f three units off the
    plane.  The reference tests each node of a searching triangle against the other triangle, both ways (Solver.cpp:757-797): the
    contact list must hold (a, b, c, d) exactly when the golden says hit, and the other five tests must come out as the fp64
    CCD of the generator decides them (cases on an edge of a triangle are left out of that part, as in the generator)."""
    gm = golden_module()
    d = load("point_triangle_ccd.npz")
    thr = float(d["threshold"])
    tune("PIES_NO_GRAPH", "1")  # 400 tiny scenes: eager launches instead of 400 graph instantiations
    hits = checked_aux = 0
    for case, (args, (hit, _t)) in enumerate(zip(d["args"], d["expected"])):
        ap0, ab0, ac0, ap1, ab1, ac1 = (v.astype(np.float32) for v in args)
        n0 = np.cross(ab0.astype(np.float64), ac0.astype(np.float64))
        n0 /= np.linalg.norm(n0)
        up = (3.0 * n0 * (1.0 if n0 @ ap0 >= 0 else -1.0))
        e = (ap0 + up + [0.4, 0.1, -0.2]).astype(np.float32)
        f = (ap0 + up + [-0.3, 0.2, 0.5]).astype(np.float32)
        zero = np.zeros(3, np.float32)
        prev = np.stack([ap0, e, f, zero, ab0, ac0])  # nodes 0..5 = a, e, f, b, c, d
        cur = np.stack([ap1, e, f, zero, ab1, ac1])
        g = pies.Solver(pies.Options(solver=pies.PD, iterations=1, gravity=0.0, floorHeight=-1.0e6, collisionThresholdDistance=thr), device=0)
        g.add_nodes_raw(cur, radius=0.01)
        g.add_triangles(np.uint32([[0, 1, 2], [3, 4, 5]]))
        g.set_prev_positions(prev)  # the swept test runs from prevPosition to position; zero velocity: the predict step moves nothing
        g.tick(1)
        got = {tuple(int(v) for v in c) for c in g.tri_collisions.reshape(-1, 4)}
        assert not g.failed
        g.close()
        assert ((0, 3, 4, 5) in got) == bool(hit), (case, hit, got)
        hits += bool(hit)
        # the other five tests of the pair, decided by the generator's fp64 CCD on the same fp32 positions
        want, sure = set(), True
        for tri, other in (((0, 1, 2), (3, 4, 5)), ((3, 4, 5), (0, 1, 2))):
            B, Cn, Dn = other
            for A in tri:
                if (A, B, Cn, Dn) == (0, 3, 4, 5):
                    continue
                rel = [(prev[A] - prev[B]), (prev[Cn] - prev[B]), (prev[Dn] - prev[B]), (cur[A] - cur[B]), (cur[Cn] - cur[B]), (cur[Dn] - cur[B])]
                ok, _, margin = gm.ccd_fp64(*rel, thr)
                if abs(margin) < 1e-3:
                    sure = False
                if ok:
                    want.add((A, B, Cn, Dn))
        if sure:
            checked_aux += 1
            assert got - {(0, 3, 4, 5)} == want, (case, got, want)
    assert 100 < hits < 300 and checked_aux > 300


# ---- whole loops -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("coll", [0, 1])
def test_pbd_tiny_scene_goldens_under_the_reference_order(pies, coll):
    """pbd_tiny_coll{0,1}.npz: an independent fp64 restatement of Solver::tickPBD (with the hash / node-node loop for coll = 1),
    one tick at a time from the stored fp32 state, schedule EXACT (the reference's container order and node-node order)"""
    d = load("pbd_tiny_coll%d.npz" % coll)
    W, Hh, D = (int(v) for v in d["dims"])
    g = pies.Solver(pies.Options(solver=pies.PBD, iterations=int(d["iterations"])), device=0)
    g.create_tet_box(W, Hh, D, translation=d["translation"], scale=float(d["spacing"]), w=float(d["w_tet"]), volume=False, triangles=False)
    g.create_box(W, Hh, D, scale=float(d["spacing"]), w=float(d["w_dist"]), existing_offset=0, triangles=False)
    g.set_radii(np.full(W * Hh * D, d["radius"], np.float32))
    g.set_flag(pies.FLAG_NODE_COLLISIONS, int(d["collisions"]))
    g.set_schedule(pies.SCHEDULE_EXACT)
    for t in range(len(d["pos"]) - 1):
        g.set_positions(d["pos"][t])
        g.set_velocities(d["vel"][t])
        g.tick(1)
        assert np.abs(g.positions - d["pos"][t + 1]).max() <= 2e-4, t
        assert np.abs(g.velocities - d["vel"][t + 1]).max() <= 2e-4 / 0.012, t
    if coll:
        assert g.collision_pairs > 0
    g.close()
