import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
def run(label, dims, iters=20, steps=40):
    g = capi.Solver(scenes.pbd_options(capi, iters)); scenes.build_beam(g, dims, distance=False); scenes.perturb(g,1,0.05); g.set_flag(1,0); g.set_schedule(capi.SCHEDULE_COLOURED); g.finalize()
    lc = sum(g.launch_counts().values())
    g.tick_async(5); g.synchronize()
    t0=time.perf_counter(); g.tick_async(steps); g.synchronize(); dt=(time.perf_counter()-t0)/steps
    print("%-28s launches/substep %5d  substep %.3f ms  => %.2f us/launch" % (label, lc, dt*1e3, dt*1e6/lc), flush=True)
    g.close()
run("tet only 100k "+os.environ.get("PIES_EXP_TET","0"), scenes.L100K)
run("tet only 1M "+os.environ.get("PIES_EXP_TET","0"), scenes.L1M, steps=10)
