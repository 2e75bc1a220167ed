#!/usr/bin/env python3
"""Dependency levels of the steady contact list of the two-box PD scene (development aid): level(c) = 1 + max level of the previous contact at each of c's four nodes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
from pies_amd import capi
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
for _ in range(30):
    g.tick_async(1); g.synchronize()
c = np.asarray(g.tri_collisions).reshape(-1, 4).astype(np.int64)
last = {}
lv = np.zeros(len(c), dtype=np.int64)
for i, ids in enumerate(c):
    l = 0
    for n in ids:
        if n in last: l = max(l, last[n] + 1)
    lv[i] = l
    for n in ids: last[n] = l
used = len(last)
cnt = np.bincount(np.concatenate([c[:, k] for k in range(4)]))
print("contacts", len(c), "touched nodes", used, "levels", lv.max() + 1, "widest level", np.bincount(lv).max(), "contacts at the busiest node", cnt.max(), "mean per touched node %.1f" % (4 * len(c) / used))
h = np.bincount(lv)
print("level widths: first 10", h[:10].tolist(), "median", int(np.median(h)), "last 10", h[-10:].tolist())
