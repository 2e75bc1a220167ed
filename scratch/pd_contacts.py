"""PD with real contacts: a short beam resting on a long one that lies on the floor."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
W,H,D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25,25,400)
D2 = max(4, D // 4)
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10))
g.create_tet_box(W,H,D, translation=(0,0.04,0), w=1.0, volume=True, triangles=True)
g.create_tet_box(W,H,D2, translation=(0.3, 0.04 + (H-1) + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
t0=time.perf_counter()
WARM, TIMED = (6, 4) if os.environ.get("PROF") else (16, 20)
for k in range(WARM):
    g.tick_async(1); g.synchronize()
    print(k, "contacts", len(g.tri_collisions), "failed", g.failed, "pcg", g.pcg_stats(), flush=True)
t0=time.perf_counter()
for k in range(TIMED): g.tick_async(1); g.synchronize()  # a host's frame loop: the CG budget follows the contacts
dt=(time.perf_counter()-t0)/TIMED
print("PD contacts %s: %.3f ms/substep %.1f substeps/s contacts %d failed %s pcg %s" % ((W,H,D), dt*1e3, 1/dt, len(g.tri_collisions), g.failed, g.pcg_stats()))
assert np.isfinite(g.positions).all()
