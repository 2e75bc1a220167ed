#!/usr/bin/env python3
"""config 3 with and without the triangle pipeline (upper bound of what a cheaper broad phase can give)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench, scenes
from pies_amd import capi
for rep in range(2):
    for tri in (True, False):
        W, H, D = scenes.L100K
        g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
        g.create_tet_box(W, H, D, translation=(0.0, 2.0, 0.0), w=1.0, volume=True, triangles=tri)
        g.add_position(np.array([D * (j + H * i) for i in range(W) for j in range(H)], dtype=np.uint32), 2.0)
        g.finalize()
        for _ in range(34):
            g.tick_async(1); g.synchronize()
        el = bench.timed_ticks(g, 30, 3, lambda: None)
        print("triangles", tri, "%.1f substeps/s (%.1f us) launches %d" % (30 / el, 1e6 * el / 30, sum(g.launch_counts().values())), flush=True)
        g.close()
