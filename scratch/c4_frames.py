"""config 4 frame by frame (tick + synchronise): per-frame time, levels, captured rounds"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
p, v = bench.config4_particles()
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p); g.set_velocities(v)
g.finalize()
tot = 0.0
for t in range(16):
    t0 = time.perf_counter(); g.tick_async(1); g.synchronize(); dt = time.perf_counter() - t0
    if t >= 2 and t < 12: tot += dt
    print("tick %2d %7.2f ms  launches/substep %5d  %s" % (t, 1e3 * dt, sum(g.launch_counts().values()), g.collision_health()), flush=True)
print("ticks 2-11 frame by frame: %.1f substeps/s" % (10 / tot))
g.close()
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p); g.set_velocities(v)
g.finalize()
el = bench.timed_ticks(g, 10, 2, lambda: None)
print("ticks 2-11 in one asynchronous call: %.1f substeps/s, launches/substep %d" % (10 / el, sum(g.launch_counts().values())))
