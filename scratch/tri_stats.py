#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
g = bench.pd_beam(scenes.L100K, 0)
g.tick(1)
print("config3", g.tri_grid_stats(), "triangles", g.count(capi.TRIANGLES), "contacts", len(g.tri_collisions))
g.close()
