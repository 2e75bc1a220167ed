import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchlib"))
import numpy as np, scenes
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import breadth_first_order
pos, tets, edges = scenes.delaunay_beam(scenes.L100K)
n = len(pos)
# BFS levels from the z-min face (multi-source): add a virtual source
e = np.concatenate([edges, edges[:, ::-1]])
src = np.nonzero(pos[:, 2] <= pos[:, 2].min() + 0.45 * np.linalg.norm(pos[edges[:, 0]] - pos[edges[:, 1]], axis=1).mean())[0]
level = np.full(n, -1); level[src] = 0
import collections
adj_ptr = np.zeros(n + 1, dtype=np.int64); np.add.at(adj_ptr, e[:, 0] + 1, 1); adj_ptr = np.cumsum(adj_ptr)
order = np.argsort(e[:, 0], kind="stable"); adj = e[order, 1]
front = list(src); L = 0
while front:
    nxt = []
    for v in front:
        for u in adj[adj_ptr[v]:adj_ptr[v + 1]]:
            if level[u] < 0: level[u] = L + 1; nxt.append(u)
    front = nxt; L += 1
print("levels", L)
tl = level[tets]; g = tl.min(1)
assert (tl.max(1) - g <= 1).all()
# per (group, node) in-group degree
key = (g[:, None] * n + tets).reshape(-1)
u, c = np.unique(key, return_counts=True)
grp = u // n
mx = np.zeros(L, dtype=int); np.maximum.at(mx, grp, c)
print("max in-group degree over all groups:", mx.max(), " mean of per-group max: %.1f" % mx.mean(), " percentiles", np.percentile(mx, [50, 90, 99]))
print("even groups max", mx[0::2].max(), "odd groups max", mx[1::2].max())
deg = np.bincount(tets.reshape(-1)); print("node degree max", deg.max())
