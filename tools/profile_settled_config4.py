#!/usr/bin/env python3
"""config 4, 12 ticks (the last four are the settled state): for a kernel trace"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
p, v = bench.config4_particles()
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p); g.set_velocities(v)
g.set_collision_rounds(64)
g.finalize()
for _ in range(12):
    g.tick_async(1); g.synchronize()
print("done", g.failed, g.collision_health())
g.close()
