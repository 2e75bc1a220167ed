import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
import oracle_api as ora
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vy = float(sys.argv[2]) if len(sys.argv) > 2 else -0.3
def build(mod, cls):
    g = cls(mod.Options(solver=mod.PD, iterations=10))
    g.create_tet_box(N, N, N, translation=(0, 0.02, 0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(N, N, N, translation=(0.4, N - 1 + 0.02 + 0.04, 0.3), w=1.0, volume=True, triangles=True)
    v = g.velocities; v[N**3:, 1] = vy; g.set_velocities(v); g.set_prev_positions(g.positions)
    return g
g = build(capi, capi.Solver); o = build(ora, ora.OracleSolver)
for t in range(60):
    g.tick(); o.tick()
    pg, po = g.positions, o.positions
    print("tick %2d gpu: contacts %4d failed %s ymin %.3f ymax %.3f | oracle: contacts %4d failed %s ymax %.3f | maxdiff %.3e" % (
        t, len(g.tri_collisions), g.failed, pg[:,1].min(), pg[:,1].max(), len(o.tri_collisions), o.failed, po[:,1].max(), np.abs(pg-po).max()))
    if g.failed or o.failed:
        print("gpu error:", g._L.pies_last_error(g._h).decode()); break
