#!/usr/bin/env python3
"""Soak runs (development aid): long sequences of ticks of the collision-heavy workloads, checking that nothing latches a
failure, everything stays finite and repeated runs agree bit for bit (the parallel collision pass is deterministic)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402


def config4(ticks):
    p, v = bench.config4_particles()
    g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
    g.addNodes(p)
    g.set_velocities(v)
    t0 = time.time()
    g.tick_async(ticks)
    g.synchronize()
    out = (g.positions.copy(), g.velocities.copy(), g.failed, g.collision_stats())
    print("config4: %d ticks in %.1f s, failed %s, pairs/candidates %s, finite %s" % (ticks, time.time() - t0, out[2], out[3], np.isfinite(out[0]).all()), flush=True)
    g.close()
    return out


a = config4(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
b = config4(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
print("two runs bit-identical:", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), flush=True)
# (w = 1 against m/h^2 = 6944 is jelly: after about a hundred ticks the beam of this scene has collapsed through itself, contacts
# run into the tens of thousands and the reference's own safety latch fires - in the oracle at tick 120 of a 1/60-size analogue;
# 40 ticks stay in the regime the parity tests cover)
g = bench.contact_scene(capi, 0)
t0 = time.time()
for f in range(40):
    g.tick()
print("contact scene: 40 synchronous ticks in %.1f s, failed %s, health %s, finite %s" % (time.time() - t0, g.failed, g.pcg_health(), np.isfinite(g.positions).all()), flush=True)
g.close()
