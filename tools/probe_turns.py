#!/usr/bin/env python3
"""Development probe of the reference-order node-node pass by turns on BASELINE config 4: per tick wall time, levels, repeats,
fallbacks and the excursions / slacks of the last pass.  usage: probe_turns.py [ticks] [iterations] [NAME=VALUE ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

args = [a for a in sys.argv[1:] if "=" not in a]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, _, v = kv.partition("=")
        capi.set_tuning(k, v)
ticks = int(args[0]) if args else 6
iters = int(args[1]) if len(args) > 1 else 4
p, v = bench.config4_particles()
if len(args) > 2 and args[2] == "settled":  # the state after 14 ticks of the default (pair) order: the burst of the over-packed block is over
    w = capi.Solver(scenes.pbd_options(capi, 4), device=0)
    w.addNodes(p)
    w.set_velocities(v)
    w.tick(14)
    p, v = w.positions, w.velocities
    w.close()
g = capi.Solver(scenes.pbd_options(capi, iters), device=0)
g.addNodes(p)
g.set_velocities(v)
g.set_flag(capi.FLAG_COLLISION_ORDER, capi.COLLISION_ORDER_REFERENCE)
g.finalize()
for t in range(ticks):
    t0 = time.perf_counter()
    g.tick_async(1)
    g.synchronize()
    dt = time.perf_counter() - t0
    slack, exc, partners = g.pair_state()
    partners = partners & 0xffff
    print("tick %2d: %8.1f ms  health %s fallbacks %d launches %d | excursion max %.3f p99 %.3f  slack max %.3f p99 %.3f mean %.3f  partners max %d mean %.1f  failed %s" % (
        t, 1e3 * dt, g.collision_health(), g.collision_fallbacks, sum(g.launch_counts().values()), exc.max(), np.quantile(exc, 0.99), slack.max(),
        np.quantile(slack, 0.99), slack.mean(), partners.max(), partners.mean(), g.failed), flush=True)
g.close()
