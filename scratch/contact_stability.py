#!/usr/bin/env python3
"""How often is the point-triangle contact list of a substep identical (same contacts, same order) to the previous substep's?  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench, scenes
from pies_amd import capi
def watch(name, g, ticks):
    prev = None; same = 0; sizes = []
    for t in range(ticks):
        g.tick_async(1); g.synchronize()
        c = np.asarray(g.tri_collisions)
        sizes.append(len(c))
        if prev is not None and c.shape == prev.shape and np.array_equal(c, prev): same += 1
        prev = c
    print(name, "contacts per substep", sizes[:5], "...", sizes[-5:], " identical to the previous substep's list: %d of %d" % (same, ticks - 1), flush=True)
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
watch("pdcontacts", g, 40)
g.close()
g = bench.contact_scene(capi, 0)
g.finalize()
watch("contacts", g, 40)
g.close()
