"""pbd 1M and the unstructured 100k beam (PBD, LAYERED): substeps/s"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
g = bench.build_scene(capi, scenes.L1M, 99, schedule=capi.SCHEDULE_LAYERED, device=0)
g.finalize()
el = bench.timed_ticks(g, 3, 1, lambda: None)
print("pbd_1m: %.1f substeps/s, launches %d" % (3 / el, sum(g.launch_counts().values())))
n, ms, units, ov = g.profile_in_situ(bench.K["layer"], 1)
print("  in situ k_layer: %d brackets avg %.2f us (overhead %.2f)" % (n, 1e3 * ms / n, 1e3 * ov))
g.close()
mesh = scenes.delaunay_beam(scenes.L100K)
g = capi.Solver(scenes.pbd_options(capi, bench.ITERATIONS), device=0)
scenes.build_unstructured(g, mesh); scenes.perturb(g, 1234, 0.03); g.set_flag(1, 0); g.set_schedule(capi.SCHEDULE_LAYERED); g.finalize()
el = bench.timed_ticks(g, 20, 2, lambda: None)
print("unstructured: %.1f substeps/s, launches %d" % (20 / el, sum(g.launch_counts().values())))
g.close()
