import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
os.environ["PIES_PCG_DEBUG"] = "1"
g = bench.pd_beam(scenes.L100K, 0)
g.finalize()
for t in range(30):
    g.tick_async(1); g.synchronize()
print("launch counts", g.launch_counts(), "health", g.pcg_health(), "stats", g.pcg_stats())
el = bench.timed_ticks(g, 20, 3, lambda: None)
print("config3: %.1f substeps/s" % (20/el))
