"""The unstructured 100k beam under PD: substeps/s and the in-situ averages of the residual kernel (13) and the CG iteration (14).
Tunables as arguments NAME=VALUE (PIES_CG_CHUNK_ROWS=512 ...): pies_set_tuning - the library does not read them from the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
for kv in sys.argv[1:]:
    capi.set_tuning(*kv.split("=", 1))
mesh = scenes.delaunay_beam(scenes.L100K)
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
scenes.build_unstructured_pd(g, mesh)
g.finalize()
for _ in range(20):
    g.tick_async(1)
    g.synchronize()
el = bench.timed_ticks(g, 20, 2, lambda: None)
res, iters, solves = g.pcg_stats()
print("unstructured pd: %.1f substeps/s, launches %d, cg iterations used %d, health %s" % (20 / el, sum(g.launch_counts().values()), iters, g.pcg_health()))
for name in ("pd_spmv", "pd_rhs", "pd_local_tet"):
    n, ms, units, ov = g.profile_in_situ(bench.K[name], 2)
    if n:
        print("  in situ %-12s: %d brackets avg %.2f us (overhead %.2f)" % (name, n, 1e3 * ms / n - 1e3 * ov, 1e3 * ov))
g.close()
