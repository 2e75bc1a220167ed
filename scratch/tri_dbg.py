#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    capi.set_tuning(k, v)
g = bench.pd_beam(scenes.L100K, 0, settle=34, pcg=(3e-7, 3))
for _ in range(8):
    g.tick_async(1); g.synchronize()
print("done", g.failed)
g.close()
