#!/usr/bin/env python3
"""Unstructured 100k beam, schedule LAYERED: slabs by position of several thicknesses / offsets against the breadth-first plan (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
mesh = scenes.delaunay_beam(scenes.L100K)
os.environ["PIES_LAYER_DEBUG"] = "1"
cases = [("1", None, None, None)] + [("2", s, o, m) for s, o, m in (
    (1.0, 0.0, None), (1.1539, 0.125, None), (1.1539, 0.375, None), (1.1539, 0.0, None), (1.077, 0.0, None),
    (1.5385, 0.5, 4000), (1.923, 0.25, 5000), (1.923, 0.625, 5000))]
for plan, slab, off, onemax in cases:
    capi.set_tuning("PIES_LAYER_PLAN", plan)
    capi.set_tuning("PIES_LAYER_PLAN_FORCE", "2" if slab else "")
    capi.set_tuning("PIES_LAYER_SLAB", str(slab) if slab else "")
    capi.set_tuning("PIES_LAYER_SLAB_OFFSET", str(off) if off else "")
    capi.set_tuning("PIES_LAYER_ONE_STRIP_MAX", str(onemax) if onemax else "")
    g = capi.Solver(scenes.pbd_options(capi, 20), device=0)
    scenes.build_unstructured(g, mesh)
    scenes.perturb(g, 1234, 0.03)
    g.set_flag(1, 0)
    g.set_schedule(capi.SCHEDULE_LAYERED)
    g.finalize()
    el = bench.timed_ticks(g, 20, 2, lambda: None)
    print("== plan", plan, "slab x", slab, "offset", off, ": %.1f substeps/s" % (20 / el), "launches", sum(g.launch_counts().values()), flush=True)
    g.close()
