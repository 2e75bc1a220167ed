#!/usr/bin/env python3
"""Generates the golden vectors in this directory (run once, fixtures are committed):

    python tests/golden/make_golden.py

Everything here is an independent float64 numpy restatement of the *maths* at the cited reference lines
(paths relative to the nithinp7/Pies tree); numpy's LAPACK SVD stands in for Eigen's JacobiSVD, which is
legitimate because only U*f(S)*V^T is consumed and that product does not depend on the SVD algorithm.
The fixtures hold inputs (float32, exactly what the oracle / the HIP path are fed) and expected outputs
(float64).  No reference source text is stored.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261003)


# ---------------------------------------------------------------------------------------------------
# single projections
# ---------------------------------------------------------------------------------------------------
def fixed_f(F, lo, hi):
    """Constraints.cpp:97-112: clamp singular values, flip the smallest if det F < 0."""
    U, S, Vt = np.linalg.svd(F)
    Sn = np.clip(S, lo, hi)
    if np.linalg.det(F) < 0:
        Sn[2] = -Sn[2]  # numpy sorts descending like Eigen
    return U @ np.diag(Sn) @ Vt


def compute_d(sigma, omin, omax):
    """Constraints.cpp:186-203."""
    D = np.zeros(3)
    for _ in range(10):
        sp = sigma + D
        prod = sp.prod()
        C = prod - np.clip(prod, omin, omax)
        g = np.array([sp[1] * sp[2], sp[0] * sp[2], sp[0] * sp[1]])
        D = (g @ D - C) * g / (g @ g)
    return D


def tet_project(x, qinv_cm, lo, hi, volume=False):
    """Constraints.cpp:76-128 / 205-255.  qinv_cm: 9 floats, column-major.  Returns the 4 projected
    vectors: (0, col0(Fhat), col1(Fhat), col2(Fhat)) with F = P*Qinv as ordinary matrices."""
    x = x.astype(np.float64)
    Qinv = qinv_cm.astype(np.float64).reshape(3, 3).T  # [col][row] -> math matrix
    P = np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]], axis=1)
    F = P @ Qinv
    if volume:
        U, S, Vt = np.linalg.svd(F)
        Fh = U @ np.diag(S + compute_d(S, lo, hi)) @ Vt
    else:
        Fh = fixed_f(F, lo, hi)
    return np.stack([np.zeros(3), Fh[:, 0], Fh[:, 1], Fh[:, 2]])


def rest_qinv(x):
    x = x.astype(np.float64)
    Q = np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]], axis=1)
    return np.linalg.inv(Q).T.reshape(9)  # column-major


def distance_project(x, target):
    """Constraints.cpp:11-37: only node a moves."""
    a, b = x.astype(np.float64)
    diff = b - a
    dist = np.linalg.norm(diff)
    d = diff / dist if dist > 1e-5 else np.array([1.0, 0, 0])
    return np.stack([a - (target - dist) * d, b])


def bend_project(x, im, angle):
    """Constraints.cpp:312-366."""
    x = x.astype(np.float64)
    im = im.astype(np.float64)
    p2, p3, p4 = x[1] - x[0], x[2] - x[0], x[3] - x[0]
    c23, c24 = np.cross(p2, p3), np.cross(p2, p4)
    l23, l24 = np.linalg.norm(c23), np.linalg.norm(c24)
    n1, n2 = c23 / l23, c24 / l24
    d = n1 @ n2
    C = np.arccos(np.clip(d, -1, 1)) - angle
    q3 = (np.cross(p2, n2) + np.cross(n1, p2) * d) / l23
    q4 = (np.cross(p2, n1) + np.cross(n2, p2) * d) / l24
    q2 = -((np.cross(p3, n2) + np.cross(n1, p3) * d) / l23) - ((np.cross(p4, n1) + np.cross(n2, p4) * d) / l24)
    q1 = -q2 - q3 - q4
    q = np.stack([q1, q2, q3, q4])
    qsq = (q * q).sum()
    out = x.copy()
    if qsq < 1e-5:
        return out
    num = np.sqrt(max(1 - d * d, 0.0)) * C
    for i in range(4):
        out[i] += -q[i] * (4 * im[i] / im.sum()) * num / qsq
    return out


def rand_rot():
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q


def gen_matrices(n):
    out = []
    for t in range(n):
        k = t % 6
        if k == 0:
            A = rng.normal(size=(3, 3))
        elif k == 1:
            A = np.eye(3) + 0.3 * rng.normal(size=(3, 3))
        elif k == 2:
            A = rand_rot() @ np.diag([1.0, 1.0 + 1e-4 * rng.normal(), 0.2]) @ rand_rot().T
        elif k == 3:
            A = rand_rot() @ np.diag([1.5, 0.9, 0.05]) @ rand_rot().T
        elif k == 4:
            A = rand_rot() @ (np.eye(3) + 0.05 * rng.normal(size=(3, 3)))
        else:  # inverted element
            A = rand_rot() @ np.diag([1.1, 0.7, -0.4]) @ rand_rot().T
        out.append(A)
    return np.array(out, dtype=np.float32)


def make_projections():
    A = gen_matrices(600)
    fixed = np.array([fixed_f(a.astype(np.float64), 0.8, 1.0) for a in A])
    np.savez(os.path.join(HERE, "svd_fixed.npz"), A=A, lo=np.float32(0.8), hi=np.float32(1.0), expected=fixed)

    n = 400
    xs, qs, exp_t, exp_v = [], [], [], []
    for t in range(n):
        rest = rng.normal(size=(4, 3)).astype(np.float32) * (0.5 + rng.uniform()) + rng.normal(size=3).astype(np.float32)
        while abs(np.linalg.det(np.stack([rest[1] - rest[0], rest[2] - rest[0], rest[3] - rest[0]]))) < 0.05:
            rest = rng.normal(size=(4, 3)).astype(np.float32)
        q = rest_qinv(rest).astype(np.float32)
        kind = t % 4
        G = [np.eye(3) + 0.1 * rng.normal(size=(3, 3)), rand_rot() @ np.diag(rng.uniform(0.5, 1.5, 3)),
             rand_rot() @ np.diag([1.0, 0.9, -0.6]), rand_rot() * rng.uniform(0.3, 2.0)][kind]
        x = ((rest.astype(np.float64) - rest[0]) @ G.T + rng.normal(size=3)).astype(np.float32)
        xs.append(x)
        qs.append(q)
        exp_t.append(tet_project(x, q, 0.8, 1.0))
        exp_v.append(tet_project(x, q, 1.0, 1.0, volume=True))
    np.savez(os.path.join(HERE, "tet_projection.npz"), x=np.array(xs), qinv=np.array(qs), lo=np.float32(0.8),
             hi=np.float32(1.0), expected=np.array(exp_t))
    np.savez(os.path.join(HERE, "volume_projection.npz"), x=np.array(xs), qinv=np.array(qs), lo=np.float32(1.0),
             hi=np.float32(1.0), expected=np.array(exp_v))

    xd = rng.normal(size=(200, 2, 3)).astype(np.float32)
    xd[0, 1] = xd[0, 0]  # coincident nodes: direction falls back to (1,0,0)
    td = rng.uniform(0.2, 2.0, 200).astype(np.float32)
    np.savez(os.path.join(HERE, "distance_projection.npz"), x=xd, target=td,
             expected=np.array([distance_project(x, t) for x, t in zip(xd, td)]))

    xb = rng.normal(size=(200, 4, 3)).astype(np.float32)
    imb = rng.uniform(0.5, 2.0, (200, 4)).astype(np.float32)
    ab = rng.uniform(0.2, 2.8, 200).astype(np.float32)
    np.savez(os.path.join(HERE, "bend_projection.npz"), x=xb, invMass=imb, angle=ab,
             expected=np.array([bend_project(x, m, a) for x, m, a in zip(xb, imb, ab)]))


# ---------------------------------------------------------------------------------------------------
# whole-loop restatements on tiny scenes (float64, plain Python loops)
# ---------------------------------------------------------------------------------------------------
OPT = dict(fixedTimestepSize=0.012, timeSubsteps=1, iterations=4, collisionStabilizationIterations=4,
           collisionThresholdDistance=0.1, collisionThickness=0.05, gravity=10.0, damping=0.006, friction=0.01,
           staticFrictionThreshold=0.0, floorHeight=0.0, gridSpacing=2.0, threadCount=8)


def lattice(W, H, D, tr, scale):
    pos = np.array([[i, j, k] for i in range(W) for j in range(H) for k in range(D)], dtype=np.float64) * scale + tr
    gid = lambda i, j, k: k + D * (j + H * i)
    tets = []
    for i in range(W - 1):
        for j in range(H - 1):
            for k in range(D - 1):
                n = {(a, b, c): gid(i + a, j + b, k + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)}
                for q in ([(0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)], [(0, 0, 0), (0, 1, 0), (0, 1, 1), (1, 1, 1)],
                          [(0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1)], [(0, 0, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1)],
                          [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1)], [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)]):
                    tets.append([n[c] for c in q])
    dist = []
    for i in range(W):
        for j in range(H):
            for k in range(D):
                if i < W - 1: dist.append([gid(i, j, k), gid(i + 1, j, k)])
                if j < H - 1: dist.append([gid(i, j, k), gid(i, j + 1, k)])
                if k < D - 1: dist.append([gid(i, j, k), gid(i, j, k + 1)])
                if i < W - 1 and j < H - 1 and k < D - 1:
                    dist.append([gid(i, j, k), gid(i + 1, j + 1, k + 1)])
                    dist.append([gid(i + 1, j, k), gid(i, j + 1, k + 1)])
                    dist.append([gid(i, j + 1, k), gid(i + 1, j, k + 1)])
                    dist.append([gid(i, j, k + 1), gid(i + 1, j + 1, k)])
    return pos, np.array(tets), np.array(dist)


def node_range(p, r, scale):
    """Solver.cpp:877-901."""
    R = (r + 0.5) / scale
    mn = p / scale - R
    lo = np.floor(mn).astype(np.int64)
    ln = np.ceil((mn - np.floor(mn)) + 2 * R).astype(np.int64)
    if (ln > 50).any():
        return lo, np.zeros(3, np.int64)
    return lo, ln


def pbd_tick(o, S, collisions):
    """Solver.cpp:40-160 (float64)."""
    dt = o["fixedTimestepSize"] / o["timeSubsteps"]
    pos, prev, vel, rad, im = S["pos"], S["prev"], S["vel"], S["radius"], S["invMass"]
    n = len(pos)
    for _ in range(o["timeSubsteps"]):
        prev[:] = pos
        pos += vel * dt + np.array([0, -o["gravity"], 0]) * dt * dt
        for _it in range(o["iterations"]):
            for (i, tgt, w) in S["position"]:
                pos[i] += w * (tgt - pos[i])
            for (a, b, target, w) in S["distance"]:
                pr = distance_project(np.stack([pos[a], pos[b]]), target)
                pos[a] += w * (pr[0] - pos[a])
                pos[b] += w * (pr[1] - pos[b])
            for (ids, q, w) in S["tet"]:
                pr = tet_project(pos[ids], q, 0.8, 1.0)
                for k in range(4):
                    pos[ids[k]] += w * (pr[k] - pos[ids[k]])
            if collisions:
                cells = {}
                for i in range(n):
                    lo, ln = node_range(pos[i], rad[i], o["gridSpacing"])
                    for dx in range(ln[0]):
                        for dy in range(ln[1]):
                            for dz in range(ln[2]):
                                cells.setdefault((lo[0] + dx, lo[1] + dy, lo[2] + dz), []).append(i)
                for i in range(n):
                    lo, ln = node_range(pos[i], rad[i], o["gridSpacing"])
                    buckets = []
                    for dx in range(ln[0]):
                        for dy in range(ln[1]):
                            for dz in range(ln[2]):
                                b = cells.get((lo[0] + dx, lo[1] + dy, lo[2] + dz))
                                if b is not None:
                                    buckets.append(b)
                    for b in buckets:
                        for j in b:
                            diff = pos[j] - pos[i]
                            dist = np.linalg.norm(diff)
                            disp = rad[i] + rad[j] - dist
                            if disp <= 0:
                                continue
                            d = diff / dist if dist > 1e-5 else np.array([1.0, 0, 0])
                            ws = im[i] + im[j]
                            pos[i] += 0.85 * -disp * d * im[i] / ws
                            pos[j] += 0.85 * disp * d * im[j] / ws
                            rv = vel[j] - vel[i]
                            pv = rv - (rv @ d) * d
                            f = 1.0 if np.linalg.norm(pv) < o["staticFrictionThreshold"] else o["friction"]
                            vel[i] += -f * pv * im[i] / ws
                            vel[j] += f * pv * im[j] / ws
            for i in range(n):
                if pos[i, 1] - rad[i] < o["floorHeight"]:
                    pos[i, 1] = o["floorHeight"] + rad[i]
        for i in range(n):
            vel[i] = (1 - o["damping"]) * (pos[i] - prev[i]) / dt
            if pos[i, 1] - rad[i] <= o["floorHeight"]:
                if np.hypot(vel[i, 0], vel[i, 2]) < 5.0:
                    vel[i, 0] = vel[i, 2] = 0
                else:
                    vel[i, 0] *= 1 - o["friction"]
                    vel[i, 2] *= 1 - o["friction"]


def tet_A(x):
    Qinv_cm = rest_qinv(x)
    Qm = Qinv_cm.reshape(3, 3)  # Qm[r][k] = Qinv[col r][row k]  (the reference's diffToBary_)
    Dm = np.array([[-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]], dtype=np.float64)
    A = np.zeros((4, 4))
    A[1:] = Qm @ Dm
    return Qinv_cm, A


def pd_tick(o, S):
    """Solver.cpp:162-486 with floor contacts only (float64, dense direct solve)."""
    h = o["fixedTimestepSize"] / o["timeSubsteps"]
    h2 = h * h
    pos, prev, vel, im = S["pos"], S["prev"], S["vel"], S["invMass"]
    n = len(pos)
    K = np.diag(1.0 / (im * h2))
    for (i, tgt, w) in S["position"]:
        K[i, i] += w
    Ad = np.array([[0.5, -0.5], [-0.5, 0.5]])
    for (a, b, target, w) in S["distance"]:
        K[np.ix_([a, b], [a, b])] += w * (Ad.T @ Ad)
    for name in ("tet", "volume"):
        for (ids, q, w, A) in S[name]:
            K[np.ix_(ids, ids)] += w * (A.T @ A)
    force_ext = np.zeros((n, 3))
    force_ext[:, 1] = -o["gravity"] / im
    for _ in range(o["timeSubsteps"]):
        pos += h * vel
        msn = pos / im[:, None] / h2
        statics = []
        T = o["threadCount"]
        tris = S["triangles"]
        for t in range(T):
            for tri in tris[t::T]:
                for i in tri:
                    if pos[i, 1] < o["floorHeight"] + o["collisionThickness"]:
                        statics.append(i)
        Ksys = K.copy()
        for i in statics:
            Ksys[i, i] += 1e4
        proj_static = {}
        for _it in range(o["iterations"]):
            f = msn.copy()
            for (i, tgt, w) in S["position"]:
                f[i] += w * tgt
            for (a, b, target, w) in S["distance"]:
                p = distance_project(np.stack([pos[a], pos[b]]), target)
                f[[a, b]] += w * (Ad.T @ Ad) @ p
            for name, lo, hi, vol in (("tet", 0.8, 1.0, False), ("volume", 1.0, 1.0, True)):
                for (ids, q, w, A) in S[name]:
                    p = tet_project(pos[ids], q, lo, hi, volume=vol)
                    f[ids] += w * (A.T @ p)
            stat_p = []
            for i in statics:
                p = pos[i].copy()
                if p[1] < 0:
                    p[1] = 0
                stat_p.append(p)
                f[i] += 1e4 * p
            pos[:] = np.linalg.solve(Ksys, f)
        for _c in range(o["collisionStabilizationIterations"]):
            for i, p in zip(statics, stat_p):
                pos[i] = p
        vel[:] = (1 - o["damping"]) * (pos - prev) / h + h * force_ext * im[:, None]
        prev[:] = pos
        for i in statics:
            pv = np.array([vel[i, 0], 0, vel[i, 2]])
            fr = 1.0 if np.linalg.norm(pv) < o["staticFrictionThreshold"] else o["friction"]
            vel[i] += -fr * pv


def make_loops():
    # ---- PBD: 4x3x3 beam.  The PBD tet projection blends world positions towards deformation-gradient
    # columns (reference quirk: Constraints.cpp:124-127 through Constraints.h:125-128), which makes the
    # dynamics strongly expanding (rounding noise grows ~100x per tick), so every tick is stored with its
    # full state and checked one tick at a time from the stored float32 state ("teacher forcing").
    W, H, D = 4, 3, 3
    for coll, tr, spacing, w_tet in ((0, np.array([0.0, 1.5, 0.0]), 1.0, 0.003), (1, np.array([0.0, 0.6, 0.0]), 0.9, 0.002)):
        pos0, tets, dist = lattice(W, H, D, tr, spacing)
        jit = rng.uniform(-0.05, 0.05, pos0.shape)
        vel0 = rng.uniform(-1, 1, pos0.shape)
        o = dict(OPT, iterations=5)
        rest = pos0.astype(np.float32).astype(np.float64)
        radius = 0.95 * 0.5 * spacing if not coll else 0.5  # with collisions: spheres overlap their lattice neighbours
        S = dict(pos=(pos0 + jit).astype(np.float32).astype(np.float64), prev=rest.copy(),
                 vel=vel0.astype(np.float32).astype(np.float64), radius=np.full(len(pos0), radius), invMass=np.ones(len(pos0)),
                 position=[],
                 distance=[(a, b, np.linalg.norm(rest[b] - rest[a]), 0.5) for a, b in dist],
                 tet=[(list(t), rest_qinv(rest[t]).astype(np.float32), w_tet) for t in tets])
        P, V = [S["pos"].astype(np.float32)], [S["vel"].astype(np.float32)]
        for _ in range(6):
            S["pos"] = P[-1].astype(np.float64)  # continue from the float32-rounded state, like the test does
            S["vel"] = V[-1].astype(np.float64)
            pbd_tick(o, S, bool(coll))
            P.append(S["pos"].astype(np.float32))
            V.append(S["vel"].astype(np.float32))
        np.savez(os.path.join(HERE, "pbd_tiny_coll%d.npz" % coll), dims=np.array([W, H, D]), translation=tr, spacing=spacing,
                 radius=np.float32(radius), iterations=5, w_dist=np.float32(0.5), w_tet=np.float32(w_tet), collisions=coll,
                 pos=np.array(P), vel=np.array(V))

    # ---- PD: 3x3x4 tet box (tets + volume, w=1), bottom layer pinned, sits near the floor ----
    W, H, D = 3, 3, 4
    pos0, tets, _ = lattice(W, H, D, np.array([0.0, 0.02, 0.0]), 1.0)
    rest = pos0.astype(np.float32).astype(np.float64)
    jit = rng.uniform(-0.03, 0.03, pos0.shape)
    vel0 = rng.uniform(-0.5, 0.5, pos0.shape)
    # surface triangles of the box, same generator order as the library (needed for the floor contacts)
    gid = lambda i, j, k: k + D * (j + H * i)
    tris = []
    for i in range(W - 1):
        for j in range(H - 1):
            tris += [[gid(i, j, 0), gid(i + 1, j + 1, 0), gid(i + 1, j, 0)], [gid(i, j, 0), gid(i, j + 1, 0), gid(i + 1, j + 1, 0)],
                     [gid(i, j, D - 1), gid(i + 1, j, D - 1), gid(i + 1, j + 1, D - 1)],
                     [gid(i, j, D - 1), gid(i + 1, j + 1, D - 1), gid(i, j + 1, D - 1)]]
    for i in range(W - 1):
        for k in range(D - 1):
            tris += [[gid(i, 0, k), gid(i + 1, 0, k), gid(i + 1, 0, k + 1)], [gid(i, 0, k), gid(i + 1, 0, k + 1), gid(i, 0, k + 1)],
                     [gid(i, H - 1, k), gid(i + 1, H - 1, k + 1), gid(i + 1, H - 1, k)],
                     [gid(i, H - 1, k), gid(i, H - 1, k + 1), gid(i + 1, H - 1, k + 1)]]
    for j in range(H - 1):
        for k in range(D - 1):
            tris += [[gid(0, j, k), gid(0, j + 1, k + 1), gid(0, j + 1, k)], [gid(0, j, k), gid(0, j, k + 1), gid(0, j + 1, k + 1)],
                     [gid(W - 1, j, k), gid(W - 1, j + 1, k), gid(W - 1, j + 1, k + 1)],
                     [gid(W - 1, j, k), gid(W - 1, j + 1, k + 1), gid(W - 1, j, k + 1)]]
    o = dict(OPT, iterations=6)
    tetrec = []
    for t in tets:
        q, A = tet_A(rest[t])
        tetrec.append((list(t), q.astype(np.float32), 1.0, A))
    pins = [gid(i, j, 0) for i in range(W) for j in range(H)]
    S = dict(pos=(pos0 + jit).astype(np.float32).astype(np.float64), prev=(pos0 + jit).astype(np.float32).astype(np.float64),
             vel=vel0.astype(np.float32).astype(np.float64), radius=np.full(len(pos0), 0.475), invMass=np.ones(len(pos0)),
             position=[(i, rest[i].copy(), 2.0) for i in pins], distance=[], tet=tetrec, volume=list(tetrec),
             triangles=np.array(tris))
    init = {k: S[k].copy() for k in ("pos", "vel")}
    traj = []
    for _ in range(4):
        pd_tick(o, S)
        traj.append(S["pos"].copy())
    np.savez(os.path.join(HERE, "pd_tiny.npz"), dims=np.array([W, H, D]), translation=np.array([0.0, 0.02, 0.0]),
             pos=init["pos"].astype(np.float32), vel=init["vel"].astype(np.float32), iterations=6, pins=np.array(pins),
             w_pin=np.float32(2.0), expected=np.array(traj))




# ---------------------------------------------------------------------------------------------------
# point-triangle CCD (CollisionDetection.cpp:227-302), float64 with numpy.roots for the cubic
# ---------------------------------------------------------------------------------------------------
def ccd_fp64(ap0, ab0, ac0, ap1, ab1, ac1, thr):
    ap0, ab0, ac0, ap1, ab1, ac1 = (np.asarray(v, np.float64) for v in (ap0, ab0, ac0, ap1, ab1, ac1))
    n0 = np.cross(ab0, ac0); n0 /= np.linalg.norm(n0)
    n1 = np.cross(ab1, ac1); n1 /= np.linalg.norm(n1)
    d0, d1 = n0 @ ap0, n1 @ ap1

    def inside(ab, ac, n, ap):
        b = np.linalg.solve(np.stack([ab, ac, n], axis=1), ap)
        return not (b[0] < 0 or b[0] > 1 or b[1] < 0 or b[1] > 1 or b[0] + b[1] > 1), b

    if d0 * d1 >= 0:
        if 0 <= d1 < thr:
            ok, b = inside(ab1, ac1, n1, ap1)
            return (ok, 0.0, min(b[0], b[1], 1 - b[0] - b[1]))
        return (False, -1.0, 1.0)
    # det[ap(t), ab(t), ac(t)] as a cubic in t
    import numpy.polynomial.polynomial as P
    def lin(a, b):
        return [np.array([a[k], b[k] - a[k]]) for k in range(3)]
    p, q, r = lin(ap0, ap1), lin(ab0, ab1), lin(ac0, ac1)
    # (polysub / polyadd, not the array operators: polymul trims trailing zero coefficients - a static point has none of degree 1 -
    # and arrays of different lengths do not subtract; for moving points the coefficients are the same numbers)
    det = P.polyadd(P.polysub(P.polymul(p[0], P.polysub(P.polymul(q[1], r[2]), P.polymul(q[2], r[1]))),
                              P.polymul(p[1], P.polysub(P.polymul(q[0], r[2]), P.polymul(q[2], r[0])))),
                    P.polymul(p[2], P.polysub(P.polymul(q[0], r[1]), P.polymul(q[1], r[0]))))
    roots = np.roots(det[::-1])
    real = sorted(x.real for x in roots if abs(x.imag) < 1e-9 and 0 <= x.real <= 1)
    if not real:
        return (False, -1.0, 1.0)
    t = real[0]
    ok, b = inside(ab0 + t * (ab1 - ab0), ac0 + t * (ac1 - ac0), None if False else
                   np.cross(ab0 + t * (ab1 - ab0), ac0 + t * (ac1 - ac0)) / np.linalg.norm(np.cross(ab0 + t * (ab1 - ab0), ac0 + t * (ac1 - ac0))),
                   ap0 + t * (ap1 - ap0))
    return (ok, t, min(b[0], b[1], 1 - b[0] - b[1]))


def make_ccd():
    rng = np.random.default_rng(777)  # own stream: independent of the fixtures generated above
    cases, exp = [], []
    while len(cases) < 400:
        b0, c0, d0 = rng.normal(size=(3, 3))
        vel = rng.normal(size=(4, 3)) * 0.3
        kind = len(cases) % 4
        n = np.cross(c0 - b0, d0 - b0); n /= np.linalg.norm(n)
        u, v = rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.3)
        foot = b0 + u * (c0 - b0) + v * (d0 - b0)
        if kind == 0:    # crosses the plane inside the triangle
            a0 = foot + 0.2 * n; vel[0] = -0.5 * n + 0.05 * rng.normal(size=3)
        elif kind == 1:  # resting within the threshold
            a0 = foot + 0.05 * n; vel *= 0.01
        elif kind == 2:  # crosses the plane outside the triangle
            a0 = b0 - 1.5 * (c0 - b0) + 0.2 * n; vel[0] = -0.5 * n
        else:            # far away
            a0 = foot + 2.0 * n
        p0 = np.stack([a0, b0, c0, d0]).astype(np.float32).astype(np.float64)
        p1 = (p0 + vel).astype(np.float32).astype(np.float64)
        args = [p0[0] - p0[1], p0[2] - p0[1], p0[3] - p0[1], p1[0] - p1[1], p1[2] - p1[1], p1[3] - p1[1]]
        hit, t, margin = ccd_fp64(*args, 0.1)
        if abs(margin) < 1e-3:
            continue  # skip decisions that sit on a triangle edge: fp32 and fp64 may legitimately differ
        cases.append(np.array(args, dtype=np.float32))
        exp.append((float(hit), t))
    np.savez(os.path.join(HERE, "point_triangle_ccd.npz"), args=np.array(cases), threshold=np.float32(0.1), expected=np.array(exp))


if __name__ == "__main__":
    make_projections()
    make_loops()
    make_ccd()
    print("golden vectors written to", HERE)
