"""BASELINE configs[2] (100k beam, PD): substeps/s (three fresh scenes) and in-situ kernel averages"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
for kv in sys.argv[1:]:
    capi.set_tuning(*kv.split("=", 1))
vals = []
for rep in range(3):
    g = bench.pd_beam(scenes.L100K, 0)
    el = bench.timed_ticks(g, 30, 3, lambda: None)
    vals.append(30 / el)
    if rep < 2:
        g.close()
print("config3: %s substeps/s, launches %d, health %s" % (" ".join("%.1f" % v for v in vals), sum(g.launch_counts().values()), g.pcg_health()))
for name in ("pd_spmv", "pd_rhs", "pd_local_tet"):
    n, ms, units, ov = g.profile_in_situ(bench.K[name], 2)
    if n:
        print("  in situ %-12s: %d brackets avg %.2f us (overhead %.2f)" % (name, n, 1e3 * ms / n - 1e3 * ov, 1e3 * ov))
g.close()
