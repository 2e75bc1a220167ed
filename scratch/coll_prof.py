import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
W,H,D = scenes.L500K
rng = np.random.default_rng(1234)
p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
p = p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]
g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
g.addNodes(p.astype(np.float32))
g.set_velocities(np.random.default_rng(4321).uniform(-1, 1, p.shape).astype(np.float32))
g.finalize()
g.tick_async(2); g.synchronize()
t0=time.perf_counter(); g.tick_async(5); g.synchronize(); dt=(time.perf_counter()-t0)/5
print("substep %.2f ms, pairs/substep %.0f failed %s" % (dt*1e3, g.collision_pairs/7, g.failed))
lc = g.launch_counts(); print(lc)
for k,name in enumerate(capi.KERNEL_NAMES[:9]):
    if lc.get(name,0)==0: continue
    n, ms, units = g.profile_substep(k)
    print("%-10s launches/substep %4d  total %.3f ms/substep  avg %.2f us" % (name, lc[name], ms/ (n/lc[name]), 1e3*ms/n))
