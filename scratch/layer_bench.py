"""Quick timing of config 2 under a schedule: python scratch/layer_bench.py [schedule] [steps]"""
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from pies_amd import capi
import scenes
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dims = tuple(int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else scenes.L100K
g = capi.Solver(scenes.pbd_options(capi, 20), device=0)
scenes.build_beam(g, dims, w_tet=float(os.environ.get('WTET', '0.05'))); scenes.perturb(g, 1234, 0.05); g.set_flag(1, 0)
g.set_schedule(sched)
t = time.time(); g.finalize(); print("finalize %.2fs" % (time.time() - t))
g.tick_async(10); g.synchronize()
t = time.perf_counter(); g.tick_async(steps); g.synchronize(); el = time.perf_counter() - t
lc = {k: v for k, v in g.launch_counts().items() if v}
print("schedule", sched, "substeps/s %.1f" % (steps / el), "ms %.3f" % (1e3 * el / steps), lc)
for k, name in enumerate(capi.KERNEL_NAMES):
    if lc.get(name):
        n, ms, units = g.profile_substep(k)
        print("  %-10s launches %4d avg %.2f us units/launch %.0f" % (name, n, 1e3 * ms / n, units / n))
p = g.positions
assert np.isfinite(p).all()
print("  extent of the body after the run: %.1f" % (p.max() - p.min()))
