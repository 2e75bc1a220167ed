#!/usr/bin/env python3
"""k_layer's time by segment kind (experimental build with PIES_EXP_LAYER_SKIP, scratch/exp/libpies_exp.so): config 2 with the
distance / tetrahedral segments skipped - results are wrong by construction, only the timing counts.  The library is built by hand,
never by build.py:
  mkdir -p scratch/exp && hipcc -c pies_amd/csrc/layer_kernels.hip -o scratch/exp/layer_kernels.hip.o -O3 -std=c++17 -fPIC \
      -ffp-contract=off -fno-fast-math -I include --offload-arch=gfx950 -DPIES_EXPERIMENTS
  hipcc -shared -o scratch/exp/libpies_exp.so $(ls pies_amd/lib/obj/*.o | grep -v layer_kernels.hip.o) scratch/exp/layer_kernels.hip.o \
      --offload-arch=gfx950"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import subprocess
if len(sys.argv) > 1:
    import numpy as np
    from pies_amd import capi
    capi.LIB_PATH = os.path.join(ROOT, "scratch", "exp", "libpies_exp.so")
    import bench, scenes
    g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
    g.finalize()
    el = bench.timed_ticks(g, 50, 5, lambda: None)
    print("skip mask %s: %.1f substeps/s, %.1f us per launch (41 launches)" % (os.environ.get("PIES_EXP_LAYER_SKIP", "0"), 50 / el, 1e6 * el / 50 / 41))
else:
    for mask in ("0", "2", "4", "6"):  # bit 1 = distance, bit 2 = tetrahedra
        env = dict(os.environ, PIES_EXP_LAYER_SKIP=mask)
        subprocess.run([sys.executable, __file__, "child"], env=env)
