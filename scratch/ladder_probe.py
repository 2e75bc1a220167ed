import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")): sys.path.insert(0, p)
import numpy as np, bench, scenes
from pies_amd import capi
t0=time.perf_counter(); g = bench.contact_scene(capi, 0); g.finalize(); print("finalize %.1f ms" % (1e3*(time.perf_counter()-t0)))
t0=time.perf_counter(); g.tick_async(1); g.synchronize(); print("first tick %.1f ms" % (1e3*(time.perf_counter()-t0)))
fr=[]
for _ in range(30):
    t0=time.perf_counter(); g.tick_async(1); g.synchronize(); fr.append(1e3*(time.perf_counter()-t0))
print("frames ms:", [round(f,2) for f in fr]); f=sorted(fr); print("median %.2f max %.2f ratio %.2f" % (f[len(f)//2], f[-1], f[-1]/f[len(f)//2]), g.pcg_health(), "failed", g.failed)
g2 = bench.pd_beam(scenes.L100K, 0, settle=0)
t0=time.perf_counter(); g2.tick_async(1); g2.synchronize(); print("config3 first tick %.1f ms" % (1e3*(time.perf_counter()-t0)))
for _ in range(34): g2.tick_async(1); g2.synchronize()
t0=time.perf_counter(); g2.tick_async(30); g2.synchronize(); print("config3 substeps/s %.1f" % (30/(time.perf_counter()-t0)), g2.pcg_health())
