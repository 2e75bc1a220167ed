import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10))
g.create_tet_box(N, N, N, translation=(0, 0.02, 0), w=1.0, volume=True, triangles=True)
g.create_tet_box(N, N, N, translation=(0.4, N - 1 + 0.02 + 0.04, 0.3), w=1.0, volume=True, triangles=True)
v = g.velocities; v[N**3:, 1] = float(sys.argv[2]) if len(sys.argv) > 2 else -2.0; g.set_velocities(v); g.set_prev_positions(g.positions)
g.finalize()
for rep in range(10):
    t0 = time.perf_counter(); g.tick_async(5); g.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("tick %2d: %.3f ms/substep  contacts %d  failed %s pcg %s" % (rep * 5, dt * 1e3, len(g.tri_collisions), g.failed, g.pcg_stats()))
