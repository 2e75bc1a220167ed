"""VERDICT r4 item 8 (development aid): how far can a better LEVELLING take the LAYERED plan of the unstructured beam?

A group's sweep needs at least as many colour steps as its busiest node has elements in the group.  With levels l(v) (adjacent
nodes differ by at most 1) an element of node v runs in group l(v) - 1 when it has a node one level down (B), in group l(v) when
it has one a level up (A), and in either when its nodes share v's level (F, free).  The chain of a sweep is
max over even groups + max over odd groups; each is at least max_v max(B_v, A_v, ceil(deg_v / 2)).
This script measures that bound for (1) the breadth-first levels the planner uses, (2) levels by position (slabs),
(3) a local search that moves nodes between levels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchlib"))
import numpy as np, scenes

pos, tets, edges = scenes.delaunay_beam(scenes.L100K)
n = len(pos)
tets = tets.astype(np.int64); edges = edges.astype(np.int64)
deg = np.bincount(tets.reshape(-1), minlength=n)
print("mesh", n, len(tets), len(edges), "tets per node mean %.1f max %d" % (deg.mean(), deg.max()))
axis = int(np.argmax(pos.max(0) - pos.min(0)))

e2 = np.concatenate([edges, edges[:, ::-1]])
order = np.argsort(e2[:, 0], kind="stable")
adj = e2[order, 1]
aptr = np.zeros(n + 1, dtype=np.int64); np.add.at(aptr, e2[:, 0] + 1, 1); aptr = np.cumsum(aptr)
tn = tets.reshape(-1)
torder = np.argsort(tn, kind="stable")
tof = torder // 4  # elements of a node, concatenated
tptr = np.zeros(n + 1, dtype=np.int64); np.add.at(tptr, tn + 1, 1); tptr = np.cumsum(tptr)


def bfs_levels():
    mean_edge = np.linalg.norm(pos[edges[:, 0]] - pos[edges[:, 1]], axis=1).mean()
    src = np.nonzero(pos[:, axis] <= pos[:, axis].min() + 0.45 * mean_edge)[0]
    level = np.full(n, -1); level[src] = 0
    front = src; L = 0
    while len(front):
        nb = np.unique(np.concatenate([adj[aptr[v]:aptr[v + 1]] for v in front]))
        nb = nb[level[nb] < 0]
        level[nb] = L + 1; front = nb; L += 1
    return level


def valid(level):
    return (np.abs(level[edges[:, 0]] - level[edges[:, 1]]) <= 1).all()


def shares(level):
    """B, A, F per node"""
    tl = level[tets]; lo = tl.min(1); hi = tl.max(1)
    B = np.zeros(n, dtype=np.int64); A = np.zeros(n, dtype=np.int64); F = np.zeros(n, dtype=np.int64)
    for k in range(4):
        v = tets[:, k]; lv = tl[:, k]
        np.add.at(B, v, (lo < lv))
        np.add.at(A, v, (lo == lv) & (hi > lv))
        np.add.at(F, v, (lo == lv) & (hi == lv))
    return B, A, F


def report(name, level):
    ok = valid(level)
    B, A, F = shares(level)
    need = np.maximum(np.maximum(B, A), (B + A + F + 1) // 2)
    L = level.max() + 1
    # per group: nodes at level g contribute max(A, .), nodes at g+1 contribute B: the relaxation per group
    gmax = np.zeros(L + 1, dtype=np.int64)
    np.maximum.at(gmax, level, np.maximum(A, (B + A + F + 1) // 2 * 0))
    np.maximum.at(gmax, np.maximum(level - 1, 0), np.where(level > 0, B, 0))
    print("%-40s valid %s levels %4d  max B %2d  max A %2d  max A+F %2d  bound per phase (forced only) even %2d odd %2d  max need %2d" % (
        name, ok, L, B.max(), A.max(), (A + F).max(), gmax[0::2].max(), gmax[1::2].max(), need.max()))
    return B, A, F


lv = bfs_levels()
B, A, F = report("breadth-first", lv)
print("   breadth-first: nodes with deg >= 40: mean B %.1f A %.1f F %.1f" % (B[deg >= 40].mean(), A[deg >= 40].mean(), F[deg >= 40].mean()))
print("   all nodes: B share %.2f A share %.2f F share %.2f" % (B.sum() / deg.sum(), A.sum() / deg.sum(), F.sum() / deg.sum()))

# levels by position
for h in (1.0, 1.3, 2.0, 2.6):
    lvl = np.floor((pos[:, axis] - pos[:, axis].min()) / h + 0.5).astype(np.int64)
    bad = np.abs(lvl[edges[:, 0]] - lvl[edges[:, 1]]) > 1
    print("slabs h=%.1f: edges spanning more than two levels: %d of %d" % (h, bad.sum(), len(edges)))
    report("slabs h=%.1f" % h, lvl)


# ---- local search from the breadth-first levels ----
def node_shares(level, v):
    ts = tof[tptr[v]:tptr[v + 1]]
    tl = level[tets[ts]]
    lo = tl.min(1); hi = tl.max(1); l = level[v]
    return int((lo < l).sum()), int(((lo == l) & (hi > l)).sum())


def search(level, tau, rounds=30, seed=1):
    rng = np.random.default_rng(seed)
    level = level.copy()
    B, A, F = shares(level)
    def pot(b, a):
        return max(0, b - tau) ** 2 + max(0, a - tau) ** 2
    for r in range(rounds):
        bad = np.nonzero(np.maximum(B, A) > tau)[0]
        if not len(bad):
            break
        moved = 0
        cand = np.unique(np.concatenate([bad] + [adj[aptr[v]:aptr[v + 1]] for v in bad]))
        rng.shuffle(cand)
        for u in cand:
            nb = adj[aptr[u]:aptr[u + 1]]
            ln = level[nb]
            touched = np.concatenate([[u], nb])
            before = sum(pot(B[w], A[w]) for w in touched)
            best = None
            for d in (-1, 1):
                nl = level[u] + d
                if nl < 0 or (np.abs(ln - nl) > 1).any():
                    continue
                old = level[u]; level[u] = nl
                sh = [node_shares(level, w) for w in touched]
                after = sum(pot(b, a) for b, a in sh)
                level[u] = old
                if after < before and (best is None or after < best[0]):
                    best = (after, nl, sh)
            if best is not None:
                level[u] = best[1]
                for w, (b, a) in zip(touched, best[2]):
                    B[w], A[w] = b, a
                moved += 1
        print("   tau %d round %d: %d bad nodes, %d moves, max B %d max A %d" % (tau, r, len(bad), moved, B.max(), A.max()), flush=True)
        if not moved:
            break
    return level


t0 = time.time()
for tau in (30, 28, 26):
    lv2 = search(lv, tau)
    report("local search tau=%d (%.0f s)" % (tau, time.time() - t0), lv2)
    np.save("/tmp/levels_tau%d.npy" % tau, lv2)
