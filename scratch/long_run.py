"""Stability sanity: 300 ticks of config 2 under each schedule; the schedules are different Gauss-Seidel orders, so the
trajectories agree physically (centre of mass, extent), not bit for bit."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from pies_amd import capi
res = {}
for name, sched in (("layered", capi.SCHEDULE_LAYERED), ("coloured", capi.SCHEDULE_COLOURED)):
    g = capi.Solver(scenes.pbd_options(capi, 20), device=0)
    scenes.build_beam(g, scenes.L100K); scenes.perturb(g, 1234, 0.05); g.set_flag(1, 0); g.set_schedule(sched)
    g.finalize()
    t = time.perf_counter(); g.tick_async(300); g.synchronize(); dt = time.perf_counter() - t
    p = g.positions
    assert np.isfinite(p).all()
    res[name] = (p.mean(0), p.min(0), p.max(0))
    print(name, "%.1f substeps/s" % (300 / dt), "com", p.mean(0), "min", p.min(0), "max", p.max(0))
d = np.abs(res["layered"][0] - res["coloured"][0]).max()
print("centre-of-mass difference between the schedules after 300 ticks: %.3g" % d)
