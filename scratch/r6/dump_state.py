import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import bench, scenes
from pies_amd import capi
g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
g.finalize()
out = {}
for t in (1, 5, 25):
    g.tick(t - (0 if not out else max(int(k[1:]) for k in out)))
    out["p%d" % t] = g.positions.copy()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "state_c2.npz"), ids=g.ids(capi.TET), rest=g.rest(capi.TET), order=g.order(capi.TET), **out)
print("saved", {k: v.shape for k, v in out.items()})
