#!/usr/bin/env python3
"""config 3 as bench.py measures it (the beam after 34 settle ticks, swinging), 8 more ticks: for a kernel trace of the moving state"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
g = bench.pd_beam(scenes.L100K, 0, settle=34, pcg=(3e-7, 3))
for _ in range(8):
    g.tick_async(1); g.synchronize()
print("done", g.failed)
g.close()
