#!/usr/bin/env python3
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
os.environ.pop("PIES_LAYER_DEBUG", None)
combos = [(None, None), (None, "512"), (None, None)]
for tile, blk in combos:
    capi.set_tuning("PIES_LAYER_TILE_NODES", tile)
    capi.set_tuning("PIES_LAYER_BLOCK", blk)
    g = bench.build_scene(capi, scenes.L1M, 99, schedule=capi.SCHEDULE_LAYERED, device=0)
    g.finalize()
    el = bench.timed_ticks(g, 10, 2, lambda: None)
    print("TILE_NODES", tile, "BLOCK", blk, "%.1f substeps/s" % (10 / el), "launches", sum(g.launch_counts().values()), flush=True)
    g.close()
