#!/usr/bin/env python3
"""A/B of PD tuning SETS on config 3 / 1M (development aid): pd_ab2.py [--scene config3|pd1m] "A=1,B=2" "A=0" ...
Each argument is one variant (comma separated NAME=VALUE, '-' = defaults).  Prints substeps/s and the largest position
difference to the first variant after the timed ticks."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

args = sys.argv[1:]
scene = "config3"
if args and args[0] == "--scene":
    scene = args[1]
    args = args[2:]
ref = None
seen = set()
for rep in range(2):
    for spec in args:
        kv = [] if spec == "-" else [x.split("=") for x in spec.split(",")]
        for name in seen:
            capi.set_tuning(name, None)
        for name, v in kv:
            capi.set_tuning(name, v)
            seen.add(name)
        dims = scenes.L100K if scene == "config3" else scenes.L1M
        g = bench.pd_beam(dims, 0, settle=int(os.environ.get("AB_SETTLE", "34")))
        T = int(os.environ.get("AB_TICKS", "30"))
        el = bench.timed_ticks(g, T, 3, lambda: None)
        res, iters, solves = g.pcg_stats()
        pos = g.positions
        if ref is None:
            ref = pos
        print("%-44s %8.1f substeps/s (%7.1f us)  launches %3d  res %.3g iters %d budget %s  maxdiff %.3g" % (
            spec, T / el, 1e6 * el / T, sum(g.launch_counts().values()), res, iters, g.pcg_health()["budget"],
            float(np.abs(pos - ref).max())), flush=True)
        g.close()
