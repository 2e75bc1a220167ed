#!/usr/bin/env python3
"""A/B of PD tunings on config 3 (development aid): pd_ab.py NAME v0 v1 ... [--scene config3|pdcontacts]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

name, vals = sys.argv[1], sys.argv[2:]
scene = "config3"
if "--scene" in vals:
    k = vals.index("--scene")
    scene = vals[k + 1]
    vals = vals[:k]
for rep in range(2):
    for v in vals:
        capi.set_tuning(name, None if v == "-" else v)
        if scene == "config3":
            g = bench.pd_beam(scenes.L100K, 0)
        elif scene == "pd1m":
            g = bench.pd_beam(scenes.L1M, 0)
        else:
            g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
            g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
            g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
        g.finalize()
        T = int(os.environ.get("AB_TICKS", "30"))
        el = bench.timed_ticks(g, T, 3, lambda: None)
        res, iters, solves = g.pcg_stats()
        print("%s=%s: %.1f substeps/s (%.1f us/substep)  launches %d  res %.3g iters %d health %s" % (
            name, v, T / el, 1e6 * el / T, sum(g.launch_counts().values()), res, iters, g.pcg_health()), flush=True)
        g.close()
