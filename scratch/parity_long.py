"""One-off: config 2 at full size, LAYERED vs the oracle replaying its order, several ticks, exact equality."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
import oracle_api as ora
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = capi.Solver(scenes.pbd_options(capi, 20), device=0); o = ora.OracleSolver(scenes.pbd_options(ora, 20))
for s in (g, o):
    scenes.build_beam(s, scenes.L100K); scenes.perturb(s, 7, 0.05); s.set_flag(1, 0)
g.set_schedule(capi.SCHEDULE_LAYERED); g.finalize()
for t in (capi.DISTANCE, capi.TET): o.permute(t, g.order(t))
for k in range(T):
    g.tick(1); t0 = time.time(); o.tick(1)
    eq = all(np.array_equal(getattr(g, n), getattr(o, n)) for n in ("positions", "velocities", "prev_positions"))
    print("tick", k, "bit-identical:", eq, "(oracle tick %.1f s)" % (time.time() - t0), flush=True)
    assert eq
