#!/usr/bin/env python3
"""PD tile plan statistics on the host (no GPU): tiles, pairs and nodes per tile, (tile, node) records per node."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np
from pies_amd import capi
import scenes
dims = tuple(int(x) for x in (sys.argv[1:4] or (20, 20, 250)))
for te in (sys.argv[4:] or ["128"]):
    capi.set_tuning("PIES_PD_TILE_ELEMS", te)
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=-1)
    g.create_tet_box(*dims, translation=(0, 0.02, 0), w=1.0, volume=True, triangles=True)
    p = g.pd_tile_plan()
    nn, ne = p["info"] & 0xffff, p["info"] >> 16
    n = dims[0] * dims[1] * dims[2]
    print("TILE_ELEMS", te, "tiles", len(nn), "pairs/tile mean %.1f max %d" % (ne.mean(), ne.max()), "nodes/tile mean %.1f" % nn.mean(),
          "records/node %.2f" % (nn.sum() / n), "full-node tiles %.2f" % (nn == 128).mean())
    g.close()
