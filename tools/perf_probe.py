#!/usr/bin/env python3
"""Quick throughput probe of one bench workload (development aid): perf_probe.py config2|config3|config4|contacts [ticks] [NAME=VALUE ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

what = sys.argv[1]
ticks = int(sys.argv[2]) if len(sys.argv) > 2 and "=" not in sys.argv[2] else 20
for kv in sys.argv[2:]:  # NAME=VALUE: tunings (pies_set_tuning)
    if "=" in kv:
        name, _, value = kv.partition("=")
        capi.set_tuning(name, value)
if what == "config2":
    g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
    classes = {"layer": 1}
elif what == "config3":
    g = bench.pd_beam(scenes.L100K, 0)
    B = bench.pd_bytes(g)
    classes = {k: B[k] for k in ("pd_local_tet", "pd_rhs", "pd_spmv", "pd_cg_update")}
elif what == "config4":
    p, v = bench.config4_particles()
    g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
    g.addNodes(p)
    g.set_velocities(v)
    classes = {"hash": 92.0, "collide": 1.0}
elif what == "contacts":
    g = bench.contact_scene(capi, 0)
    g.finalize()
    bench.frame_loop(g, 12)
    B = bench.pd_bytes(g)
    classes = {k: B[k] for k in ("pd_local_tet", "pd_rhs", "pd_spmv")}
elif what == "pdcontacts":
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
    g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
    g.finalize()
    for _ in range(20):
        t0 = time.perf_counter()
        g.tick_async(1)
        g.synchronize()
        print("frame %.3f ms  contacts %d  health %s" % (1e3 * (time.perf_counter() - t0), len(g.tri_collisions), g.pcg_health()))
    B = bench.pd_bytes(g)
    classes = {k: B[k] for k in ("pd_local_tet", "pd_rhs", "pd_spmv", "pd_cg_update")}
if os.environ.get("PROBE_NO_TRI") == "1":
    g.set_flag(capi.FLAG_TRIANGLE_COLLISIONS, 0)
g.finalize()
el = bench.timed_ticks(g, ticks, 3, lambda: None)
print("%s: %.1f substeps/s (%.3f ms/substep), %d launches/substep, failed %s" % (what, ticks / el, 1e3 * el / ticks, sum(g.launch_counts().values()), g.failed))
for cls, per in classes.items():
    n, ms, units, ov = g.profile_in_situ(bench.K[cls], 2)
    if n:
        net = max(ms - n * ov, 0.05 * ms)
        print("  in situ %-14s %4d brackets  avg %9.1f us (bracket %.1f - overhead %.1f)   %8.1f GB/s by the survey's bytes" % (
            cls, n, 1e3 * net / n, 1e3 * ms / n, 1e3 * ov, per * units / (net * 1e-3) / 1e9))
    try:
        n, ms, units = g.profile_substep(bench.K[cls])
        if n:
            print("  replay  %-14s %4d launches  avg %9.1f us" % (cls, n, 1e3 * ms / n))
    except Exception as e:  # noqa: BLE001
        print("  replay", cls, "failed:", e)
g.close()
