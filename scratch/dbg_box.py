import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
dims = (20, 20, 250)
g = capi.Solver(scenes.pbd_options(capi, 20), device=0)
scenes.build_beam(g, dims, tets=False); scenes.perturb(g, 1234, 0.05); g.set_flag(capi.FLAG_NODE_COLLISIONS, 1)
g.finalize()
print("radius", g.radii[:3] if hasattr(g,'radii') else None, "launches", sum(g.launch_counts().values()), g.launch_counts())
for t in range(8):
    t0 = time.perf_counter(); g.tick_async(1); g.synchronize(); dt = time.perf_counter() - t0
    h = g.collision_health(); sl, ex, dg = g.pair_state(); fin = np.isfinite(sl)
    p = g.positions
    print("tick", t, "%.1f ms" % (1e3 * dt), h, "failed", g.failed, "slack q50/max %.3f %.3f exc q50/q99/max %.3f %.3f %.3f deg mean/max %.1f %d" % (
        np.quantile(sl[fin], .5), sl[fin].max(), np.quantile(ex, .5), np.quantile(ex, .99), ex.max(), dg.mean(), dg.max()), "bbox", p.min(0), p.max(0), flush=True)
    if g.failed: print(g.last_error()); break
