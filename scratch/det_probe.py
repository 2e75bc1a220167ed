"""run config 4 twice, compare positions bitwise after every tick"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p, v = bench.config4_particles()
def make():
    g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
    g.addNodes(p); g.set_velocities(v); g.set_collision_rounds(rounds); g.finalize(); return g
a, b = make(), make()
for t in range(ticks):
    a.tick(1); b.tick(1)
    pa, pb = a.positions, b.positions
    print("tick", t, "equal", np.array_equal(pa, pb), "max diff", float(np.abs(pa - pb).max()), "pairs", a.collision_pairs, b.collision_pairs, a.collision_health(), b.collision_health(), flush=True)
