import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
dims = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else scenes.L100K
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cg = int(sys.argv[5]) if len(sys.argv) > 5 else 12
w = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
W,H,D = dims
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10)); g.set_pcg(3e-7, cg)
g.create_tet_box(W,H,D, translation=(0,2.0,0), w=w, volume=True, triangles=True)
g.add_position(np.array([D*(j+H*i) for i in range(W) for j in range(H)], dtype=np.uint32), 2.0 * w)
g.set_flag(capi.FLAG_TRIANGLE_COLLISIONS, int(os.environ.get('TRI','1'))); g.finalize()
for _ in range(34): g.tick_async(1); g.synchronize()
t0=time.perf_counter(); g.tick_async(steps); g.synchronize(); dt=(time.perf_counter()-t0)/steps
print("PD %s: %.3f ms/substep %.1f substeps/s pcg %s" % (dims, dt*1e3, 1/dt, g.pcg_stats()))
