#!/usr/bin/env python3
"""Development probe of BASELINE config 4 in the default (pair) order: the burst window (ticks 2-11 in one asynchronous call) and ten
settled frames (ticks 12-21, one tick + one synchronisation each), as bench.py measures them.  usage: probe_config4.py [NAME=VALUE ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

for kv in sys.argv[1:]:
    if "=" in kv:
        k, _, v = kv.partition("=")
        capi.set_tuning(k, v)
p, v = bench.config4_particles()
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p)
g.set_velocities(v)
if os.environ.get("PROBE_ROUNDS"):
    g.set_collision_rounds(int(os.environ["PROBE_ROUNDS"]))  # (pins the captured level launches)
g.finalize()
g.tick_async(2)
g.synchronize()
burst = 10 / bench.timed_ticks(g, 10, 0, lambda: None)
frames = []
for _ in range(10):
    t0 = time.perf_counter()
    g.tick_async(1)
    g.synchronize()
    frames.append(time.perf_counter() - t0)
print("config 4 %s: burst %.1f substeps/s, settled %.1f (frames ms: %s)  health %s fallbacks %d launches %d failed %s" % (
    " ".join(a for a in sys.argv[1:]), burst, 10 / sum(frames), " ".join("%.2f" % (1e3 * f) for f in frames), g.collision_health(),
    g.collision_fallbacks, sum(g.launch_counts().values()), g.failed), flush=True)
g.close()
