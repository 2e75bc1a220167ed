"""forced-share bound (level_search.py) of slab levellings over thickness x offset (tets and distance constraints)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchlib"))
import numpy as np, scenes
pos, tets, edges = scenes.delaunay_beam(scenes.L100K)
n = len(pos); tets = tets.astype(np.int64); edges = edges.astype(np.int64)
axis = int(np.argmax(pos.max(0) - pos.min(0)))
z = (pos[:, axis] - pos[:, axis].min()).astype(np.float64)
print("longest z extent of a tet %.3f" % (z[tets].max(1) - z[tets].min(1)).max())
def forced(level, ops):
    tl = level[ops]; lo = tl.min(1); hi = tl.max(1)
    if (hi - lo > 1).any(): return None
    low = np.zeros(n, dtype=np.int64); high = np.zeros(n, dtype=np.int64)
    for k in range(ops.shape[1]):
        v = ops[:, k]; lv = tl[:, k]
        np.add.at(low, v, lo < lv); np.add.at(high, v, (lo == lv) & (hi > lv))
    L = level.max() + 1
    gm = np.zeros(L + 1, dtype=np.int64)
    np.maximum.at(gm, level, high); np.maximum.at(gm, np.maximum(level - 1, 0), np.where(level > 0, low, 0))
    return gm[0::2].max(), gm[1::2].max()
print("rows: thickness; columns: offset 0, 1/8 .. 7/8; entries: tet even+odd forced bound (distance bound)")
for h in (2.6, 2.8, 3.0, 3.2, 3.5, 4.0, 4.5, 5.0, 6.0):
    row = []
    for o in np.arange(8) / 8.0:
        level = np.floor(z / h + o).astype(np.int64)
        f = forced(level, tets); d = forced(level, edges[:, :1].repeat(2, 1) * 0 + edges)
        row.append("  -  " if f is None else "%2d+%2d" % f)
    print("h=%.1f: " % h + "  ".join(row), flush=True)
