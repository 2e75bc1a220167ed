#!/usr/bin/env python3
"""host-only: time of the LAYERED plan and its colour counts (PIES_LAYER_DEBUG=1) for the lattice and the Delaunay beam"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
os.environ["PIES_LAYER_DEBUG"] = "1"
import numpy as np
import scenes
from pies_amd import capi
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    capi.set_tuning(k, v)
which = sys.argv[1] if len(sys.argv) > 1 else "lattice"
g = capi.Solver(scenes.pbd_options(capi, 20), device=-1)
if which == "lattice":
    scenes.build_beam(g, scenes.L100K)
else:
    scenes.build_unstructured(g, scenes.delaunay_beam(scenes.L100K))
g.set_flag(capi.FLAG_NODE_COLLISIONS, 0)
g.set_schedule(capi.SCHEDULE_LAYERED)
t0 = time.perf_counter()
o = g.order(capi.TET)
print("plan time %.2f s, order length %d" % (time.perf_counter() - t0, len(o)))
