import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
import scenes
from pies_amd import capi
os.environ["PIES_LAYER_DEBUG"] = "1"
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    capi.set_tuning(k, v)
mesh = scenes.delaunay_beam(scenes.L100K)
print("mesh", len(mesh[0]), len(mesh[1]), len(mesh[2]))
deg = np.bincount(mesh[1].reshape(-1)); print("tets per node: mean %.1f max %d" % (deg.mean(), deg.max()))
g = capi.Solver(scenes.pbd_options(capi, 20), device=capi.DEVICE_NONE)
scenes.build_unstructured(g, mesh)
g.set_flag(1, 0)
g.set_schedule(capi.SCHEDULE_LAYERED)
t = time.time(); g.finalize(); print("finalize %.1f s" % (time.time() - t))
