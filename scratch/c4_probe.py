"""config 4 probe: substeps/s and health for a collision order"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 96
p, v = bench.config4_particles()
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p); g.set_velocities(v)
g.set_flag(capi.FLAG_COLLISION_ORDER, order)
g.set_collision_rounds(rounds) if rounds else None
g.finalize()
for t in range(6):
    t0 = time.perf_counter(); g.tick_async(1); g.synchronize(); print("tick", t, "%.1f ms" % (1e3 * (time.perf_counter() - t0)), g.collision_health(), flush=True)
    sl, ex, dg = g.pair_state()
    fin = np.isfinite(sl)
    print("   slack q50/q99/max %.3f %.3f %.3f  exc q50/q99/max %.3f %.3f %.3f  deg mean/max %.1f %d  inf-slack %d" % (
        np.quantile(sl[fin], .5), np.quantile(sl[fin], .99), sl[fin].max(), np.quantile(ex, .5), np.quantile(ex, .99), ex.max(), dg.mean(), dg.max(), (~fin).sum()), flush=True)
g.collision_stats()
print("health after 2 ticks", g.collision_health(), "failed", g.failed, flush=True)
t0 = time.perf_counter(); g.tick_async(10); g.synchronize(); el = time.perf_counter() - t0
pairs, cand = g.collision_stats()
print("order", order, "rounds", rounds, "substeps/s %.2f" % (10 / el), "pairs/substep", pairs / 10, "cand/node/iter", cand / (40 * len(p)), g.collision_health(), "launches", sum(g.launch_counts().values()), flush=True)
for k in ("hash", "collide"):
    l, ms, u, oh = g.profile_in_situ(bench.K[k], 1)
    print(k, "launches", l, "ms total %.3f" % ms, "overhead/launch us %.2f" % (1e3 * oh), flush=True)
