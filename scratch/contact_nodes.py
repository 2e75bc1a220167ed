#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench
from pies_amd import capi
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
for t in range(8):
    g.tick()
    c = g.tri_collisions
    print("pdcontacts tick", t, "contacts", len(c), "nodes", len(np.unique(c)), g.tri_grid_stats())
g.close()
g = bench.contact_scene(capi, 0)
g.finalize()
for t in range(6):
    g.tick()
    c = g.tri_collisions
    print("config5 tick", t, "contacts", len(c), "nodes", len(np.unique(c)))
