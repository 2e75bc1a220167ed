"""VERDICT r4 item 8 (development aid): the colour steps a LAYERED plan of the Delaunay beam needs with levels by position
(slabs of thickness h, offset o) once the single-level ("free") elements are dealt CONSISTENTLY (an element goes to one group
with all four nodes) - the relaxation in level_search.py deals per node.  Greedy dealing + flipping passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchlib"))
import numpy as np, scenes

pos, tets, edges = scenes.delaunay_beam(scenes.L100K)
n = len(pos); tets = tets.astype(np.int64); edges = edges.astype(np.int64)
axis = int(np.argmax(pos.max(0) - pos.min(0)))
z = pos[:, axis] - pos[:, axis].min()


def deal(level, passes=6, seed=0):
    tl = level[tets]; lo = tl.min(1); hi = tl.max(1)
    ok = hi - lo <= 1
    low = np.zeros(n, dtype=np.int64); high = np.zeros(n, dtype=np.int64)  # a node's load in group level-1 / in group level
    for k in range(4):
        v = tets[:, k]; lv = tl[:, k]
        np.add.at(low, v, ok & (lo < lv))
        np.add.at(high, v, ok & (lo == lv) & (hi > lv))
    forced = max(low.max(), high.max())
    free = np.nonzero(ok & (hi == lo))[0]
    rng = np.random.default_rng(seed); rng.shuffle(free)
    side = np.zeros(len(tets), dtype=np.int8)
    T = tets
    for e in free:
        vs = T[e]
        a = low[vs].max(); b = high[vs].max()
        if a <= b: low[vs] += 1; side[e] = 1
        else: high[vs] += 1; side[e] = 2
    hist = [max(low.max(), high.max())]
    for p in range(passes):
        flips = 0
        for e in free:
            vs = T[e]
            if side[e] == 1:
                cur = low[vs].max(); oth = high[vs].max() + 1
                if oth < cur: low[vs] -= 1; high[vs] += 1; side[e] = 2; flips += 1
            else:
                cur = high[vs].max(); oth = low[vs].max() + 1
                if oth < cur: high[vs] -= 1; low[vs] += 1; side[e] = 1; flips += 1
        hist.append(max(low.max(), high.max()))
        if not flips: break
    # per phase: group g's busiest node = max(high of nodes at level g, low of nodes at level g+1)
    L = level.max() + 1
    gm = np.zeros(L + 1, dtype=np.int64)
    np.maximum.at(gm, level, high); np.maximum.at(gm, np.maximum(level - 1, 0), np.where(level > 0, low, 0))
    return forced, hist, gm[0::2].max(), gm[1::2].max(), int((~ok).sum()), len(free)


for h, o in ((2.6, 0.5), (2.0, 0.5), (2.0, 0.25), (2.0, 0.0), (2.0, 0.75), (2.2, 0.5), (2.4, 0.5), (3.0, 0.5), (3.0, 0.0), (4.0, 0.5)):
    t0 = time.time()
    level = np.floor(z / h + o).astype(np.int64)
    forced, hist, ev, od, bad, nfree = deal(level)
    print("slabs h=%.1f offset %.2f: %3d levels, %5d elements over three levels (ignored), %6d free; busiest node forced %2d, dealt %s; "
          "phases %2d + %2d = %d colour steps at best  (%.0f s)" % (h, o, level.max() + 1, bad, nfree, forced, hist, ev, od, ev + od, time.time() - t0), flush=True)
