#!/usr/bin/env python3
"""The program tools/profile_round.sh puts under rocprofv3: N ticks of one of bench.py's workloads, whole substeps
(every kernel in its place), nothing else.  usage: profile_target.py config2|config3|config4|contacts|pdcontacts|pbd1m|pd1m|pd1m_work|pd1m_streamed|pd_unstructured|pd_unstructured_1m N [NAME=VALUE ...]
(the *_work / *_streamed / pd_unstructured* targets run every captured CG launch as a working iteration - PIES_PCG_NEVER_EXIT: a body at
rest converges at its first look at the residual, and a trace of it holds nothing but early exits; pd1m_streamed switches the row
dictionary off so that the 100^3 lattice streams its matrix like an unstructured mesh does)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

what, n = sys.argv[1], int(sys.argv[2])
for kv in sys.argv[3:]:  # further arguments: NAME=VALUE tunings (pies_set_tuning), e.g. PIES_PD_FUSE_RHS=0
    name, _, value = kv.partition("=")
    capi.set_tuning(name, value)
if what in ("contacts", "pdcontacts"):
    # PIES_PROFILER_SAFE freezes the graph variant as well: take the one bench.py's loop runs these scenes in (contact rows
    # summed by the extra blocks of k_cg_ap, level kernel on the second stream)
    capi.set_tuning("PIES_TRI_FAST_ROWS", "1")
if what == "config2":
    g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
elif what == "config3":
    # PIES_PROFILER_SAFE keeps the captured CG budget where it starts: 3 = what bench.py's loop settles at
    g = bench.pd_beam(scenes.L100K, 0, settle=0, pcg=(3e-7, 3))
elif what == "pbd1m":
    g = bench.build_scene(capi, scenes.L1M, 99, schedule=capi.SCHEDULE_LAYERED, device=0)
elif what == "pd1m":
    g = bench.pd_beam(scenes.L1M, 0, settle=0, pcg=(3e-7, 3))
elif what in ("pd1m_work", "pd1m_streamed"):
    capi.set_tuning("PIES_PCG_NEVER_EXIT", "1")
    if what == "pd1m_streamed":
        capi.set_tuning("PIES_PD_ROW_DICT", "0")
    g = bench.pd_beam(scenes.L1M, 0, settle=0, pcg=(3e-7, 3))
elif what in ("pd_unstructured", "pd_unstructured_1m"):
    capi.set_tuning("PIES_PCG_NEVER_EXIT", "1")
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
    scenes.build_unstructured_pd(g, scenes.delaunay_beam(scenes.L1M if what.endswith("_1m") else scenes.L100K))
    g.set_pcg(3e-7, 3)
elif what == "config4":
    p, v = bench.config4_particles()
    g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
    g.addNodes(p)
    g.set_velocities(v)
elif what == "contacts":
    g = bench.contact_scene(capi, 0)
    g.set_pcg(3e-7, 12)  # (see config3; bench.py's frame loop settles at about 10 with these contacts)
    g.finalize()  # (the profiled ticks are the contact onset: frames 0 .. n-1, thousands of contacts in frames 0-3)
elif what == "pdcontacts":
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
    g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
    g.set_pcg(3e-7, 10)  # (bench.py's frame loop settles at 10 in this scene)
else:
    raise SystemExit("unknown workload")
g.finalize()
for _ in range(n):
    g.tick_async(1)
    g.synchronize()
print(what, "ticks", n, "failed", g.failed)
g.close()
