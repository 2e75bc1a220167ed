import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
os.environ["PIES_LAYER_DEBUG"] = "1"
import scenes
from pies_amd import capi
g = capi.Solver(scenes.pbd_options(capi, 20), device=capi.DEVICE_NONE)
scenes.build_beam(g, scenes.L100K); g.set_flag(1, 0); g.set_schedule(capi.SCHEDULE_EXACT); g.finalize()
