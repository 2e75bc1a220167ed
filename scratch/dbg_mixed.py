import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, scenes
from pies_amd import capi as pies
import oracle_api as oracle
def build(s):
    scenes.build_beam(s, (7, 6, 13), translation=(0.0, 0.3, 0.0))
    scenes.build_beam(s, (5, 9, 4), translation=(1.3, 7.2, 0.4))
    s.create_bend_sheet(7, 9, translation=(12.0, 3.0, 0.0))
    s.create_sheet(9, 7, translation=(24, 3, 0), scale=0.5, mass=2.0, w=0.7)
    s.add_position(np.array([3, 3, 40], dtype=np.uint32), 0.3)
    s.addNodes(np.float32([[40, 0.4, 0], [41, 6, 1], [42, 0.2, 2], [43, 3, 3], [44, 0.1, 4]]))
    scenes.perturb(s, 4, 0.05)
for iters, ticks in ((1, 1), (4, 1), (4, 3)):
  for order, rule in ((pies.COLLISION_ORDER_PAIRS, 2), (pies.COLLISION_ORDER_GROUPS, 1)):
    g = pies.Solver(scenes.pbd_options(pies, iters), device=0)
    o = oracle.OracleSolver(scenes.pbd_options(oracle, iters))
    for s in (g, o):
        build(s); s.set_flag(1, 1)
    o.set_flag(oracle.FLAG_COLLISION_RULE, rule)
    g.set_schedule(pies.SCHEDULE_COLOURED); g.set_flag(pies.FLAG_COLLISION_ORDER, order)
    g.finalize()
    for t in (pies.DISTANCE, pies.TET, pies.BEND):
        o.permute(t, g.order(t))
    g.tick(ticks); o.tick(ticks)
    d = np.abs(g.positions - o.positions).max(1)
    bad = np.nonzero(d > 1e-5)[0]
    print("iters", iters, "ticks", ticks, "order", order, "max", d.max(), "nbad", len(bad), "first bad", bad[:10], "pairs", g.collision_pairs, o.collision_pairs, g.collision_health(), "failed", g.failed, flush=True)
    if len(bad):
        i = bad[0]; print("   node", i, "radius", g.radii[i] if hasattr(g, "radii") else None, "pos", g.positions[i], o.positions[i])
