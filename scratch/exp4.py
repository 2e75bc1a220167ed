import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
from test_collisions_gpu import particles
def timeit(g, steps, label, extra=""):
    g.finalize(); g.tick_async(2); g.synchronize()
    t0=time.perf_counter(); g.tick_async(steps); g.synchronize(); dt=(time.perf_counter()-t0)/steps
    print("%-34s %.3f ms/substep  %.1f substeps/s %s" % (label, dt*1e3, 1/dt, extra), flush=True)
p,v = particles((40,50,50))
g = capi.Solver(scenes.pbd_options(capi, 4)); g.addNodes(p); g.set_velocities(v)
timeit(g, 10, "100k loose particles 4 it"); print("pairs", g.collision_pairs, "failed", g.failed); g.close()
p,v = particles(scenes.L500K)
g = capi.Solver(scenes.pbd_options(capi, 4)); g.addNodes(p); g.set_velocities(v)
timeit(g, 5, "config4 500k collisions 4 it"); print("pairs", g.collision_pairs, "failed", g.failed); g.close()
# spacing 1.0 (touching, few overlaps): broadphase-dominated
p,v = particles(scenes.L500K, spacing=1.0, jitter=0.02)
g = capi.Solver(scenes.pbd_options(capi, 4)); g.addNodes(p); g.set_velocities(v)
timeit(g, 5, "500k spacing 1.0 (sparse contacts)"); print("pairs", g.collision_pairs, "failed", g.failed); g.close()
