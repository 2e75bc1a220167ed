#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
for dims in ((70, 70, 70), (50, 50, 200), (30, 30, 400)):
    for blk in (None, "512", "256"):
        capi.set_tuning("PIES_LAYER_BLOCK", blk)
        g = bench.build_scene(capi, dims, 7, schedule=capi.SCHEDULE_LAYERED, device=0)
        g.finalize()
        el = bench.timed_ticks(g, 10, 2, lambda: None)
        print(dims, "PIES_LAYER_BLOCK", blk, "%.1f substeps/s" % (10 / el), "layer launches", g.launch_counts()["layer"], flush=True)
        g.close()
