#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
mesh = scenes.delaunay_beam(scenes.L100K)
for rep in range(2):
    for plan in ("0", "1", "2"):
        capi.set_tuning("PIES_LAYER_PLAN", plan)
        g = capi.Solver(scenes.pbd_options(capi, 20), device=0)
        scenes.build_unstructured(g, mesh)
        scenes.perturb(g, 1234, 0.03)
        g.set_flag(1, 0)
        g.set_schedule(capi.SCHEDULE_LAYERED)
        g.finalize()
        el = bench.timed_ticks(g, 20, 2, lambda: None)
        print("PIES_LAYER_PLAN", plan, "%.1f substeps/s" % (20 / el), "launches", sum(g.launch_counts().values()), flush=True)
        g.close()
