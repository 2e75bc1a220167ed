#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
for rep in range(2):
    for blk in (None, "256", "512", "1024"):
        capi.set_tuning("PIES_LAYER_BLOCK", blk)
        g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
        g.finalize()
        el = bench.timed_ticks(g, 50, 5, lambda: None)
        print("PIES_LAYER_BLOCK", blk, "%.1f substeps/s" % (50 / el), flush=True)
        g.close()
