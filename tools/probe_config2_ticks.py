"""config 2 tick by tick: ms per tick (41 launches of k_layer) as the beam collapses onto the floor."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench, scenes
from pies_amd import capi
g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
g.finalize()
for t in range(45):
    t0 = time.perf_counter()
    g.tick_async(1); g.synchronize()
    dt = time.perf_counter() - t0
    if t < 6 or t % 5 == 0:
        p = g.positions
        print("tick %2d: %.3f ms   bbox y %.3f..%.3f  x %.2f..%.2f" % (t, 1e3 * dt, p[:,1].min(), p[:,1].max(), p[:,0].min(), p[:,0].max()), flush=True)
el = bench.timed_ticks(g, 20, 3, lambda: None)
print("steady: %.1f substeps/s (%.3f ms/substep)" % (20 / el, 1e3 * el / 20))
n, ms, units, ov = g.profile_in_situ(bench.K["layer"], 2)
print("in situ k_layer: %d brackets avg %.2f us (overhead %.2f)" % (n, 1e3 * ms / n, 1e3 * ov))
