#!/usr/bin/env python3
"""pies_tick (synchronous, positions on the host afterwards) against pies_tick_async on config 2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench, scenes
from pies_amd import capi
g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
g.finalize()
g.tick(5)
for rep in range(3):
  for mode in ("1", "0"):
    capi.set_tuning("PIES_READBACK_PACK", mode)
    t0 = time.perf_counter()
    for _ in range(100): g.tick()
    t1 = time.perf_counter()
    for _ in range(100):
        g.tick(); g.read_positions_strided(9)
    t1b = time.perf_counter()
    g.tick_async(100); g.synchronize()
    t2 = time.perf_counter()
    print("pack", mode, "pies_tick %.1f /s  + strided write %.1f /s   tick_async %.1f /s   difference %.1f us per tick" % (100 / (t1 - t0), 100 / (t1b - t1), 100 / (t2 - t1b), 1e4 * ((t1 - t0) - (t2 - t1b))))
p = g.positions
g.tick_async(1); g.synchronize()
print("finite", np.isfinite(g.positions).all())
