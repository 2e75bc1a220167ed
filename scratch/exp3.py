import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
from test_collisions_gpu import particles
from test_pd_parity_gpu import build_pd_beam, pd_options
def timeit(g, steps, label, extra=""):
    g.finalize(); g.tick_async(3); g.synchronize()
    t0=time.perf_counter(); g.tick_async(steps); g.synchronize(); dt=(time.perf_counter()-t0)/steps
    print("%-34s %.3f ms/substep  %.1f substeps/s %s" % (label, dt*1e3, 1/dt, extra), flush=True)
# config 4
p,v = particles(scenes.L500K)
g = capi.Solver(scenes.pbd_options(capi, 4)); g.addNodes(p); g.set_velocities(v)
timeit(g, 10, "config4 500k collisions 4 it", str(g.launch_counts()))
print("pairs/10 ticks", g.collision_pairs); g.close()
# 100k particles collisions
p,v = particles((40,50,50))
g = capi.Solver(scenes.pbd_options(capi, 4)); g.addNodes(p); g.set_velocities(v)
timeit(g, 20, "100k loose particles 4 it"); g.close()
# config 3 PD
g = capi.Solver(pd_options(capi, 10)); build_pd_beam(g, scenes.L100K, translation=(0,2.0,0))
timeit(g, 20, "config3 PD 100k 10 it", ""); print("pcg", g.pcg_stats()); g.close()
g = capi.Solver(pd_options(capi, 10)); g.set_pcg(3e-7, 6); build_pd_beam(g, scenes.L100K, translation=(0,2.0,0))
timeit(g, 20, "config3 PD 100k 10 it, 6 cg", ""); print("pcg", g.pcg_stats()); g.close()
# config 2 with reference-default collisions on
g = capi.Solver(scenes.pbd_options(capi, 20)); scenes.build_beam(g, scenes.L100K); scenes.perturb(g,1,0.05); g.set_schedule(1)
timeit(g, 5, "config2 + node collisions (ref default)"); g.close()
