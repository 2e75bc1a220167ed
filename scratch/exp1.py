import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, scenes
from pies_amd import capi
def run(label, build, iters=20, steps=50):
    g = capi.Solver(scenes.pbd_options(capi, iters)); build(g); g.set_flag(1,0); g.set_schedule(capi.SCHEDULE_COLOURED); g.finalize()
    lc = sum(g.launch_counts().values())
    g.tick_async(5); g.synchronize()
    t0=time.perf_counter(); g.tick_async(steps); g.synchronize(); dt=(time.perf_counter()-t0)/steps
    print("%-28s launches/substep %5d  substep %.3f ms  => %.2f us/launch" % (label, lc, dt*1e3, dt*1e6/lc), flush=True)
    g.close()
dims = scenes.L100K
def nodes_only(g): g.create_tet_box(*dims, translation=(0,5,0), w=0.05); 
import ctypes
run("beam dist+tet", lambda g: (scenes.build_beam(g, dims), scenes.perturb(g,1,0.05)))
run("dist only", lambda g: (scenes.build_beam(g, dims, tets=False), scenes.perturb(g,1,0.05)))
run("tet only", lambda g: (scenes.build_beam(g, dims, distance=False), scenes.perturb(g,1,0.05)))
def just_nodes(g):
    W,H,D=dims
    import itertools
    p=np.stack(np.meshgrid(np.arange(W),np.arange(H),np.arange(D),indexing='ij'),-1).reshape(-1,3).astype(np.float32)+np.float32([0,5,0])
    g.addNodes(p)
run("nodes only (floor x20)", just_nodes)
run("1M tet only", lambda g: (scenes.build_beam(g, scenes.L1M, distance=False), scenes.perturb(g,1,0.05)), steps=10)
