#!/usr/bin/env python3
"""Development probe of the PD global step's kernels (k_cg1_init / k_cg1_first / k_cg1_iter) with the system matrix streamed
or not: probe_cg.py pd1m|pd1m_streamed|pd100k|pd100k_streamed|unstructured|unstructured_1m [NAME=VALUE ...]
Prints whole-substep throughput and the in-situ time of the CG iteration (converged exit off) and of the residual kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402


def make(what):
    if what.endswith("_streamed"):
        capi.set_tuning("PIES_PD_ROW_DICT", "0")
        what = what[:-len("_streamed")]
    else:
        capi.set_tuning("PIES_PD_ROW_DICT", "1")
    if what == "pd1m":
        return bench.pd_beam(scenes.L1M, 0, settle=6)
    if what == "pd100k":
        return bench.pd_beam(scenes.L100K, 0, settle=12)
    if what in ("unstructured", "unstructured_1m"):
        mesh = scenes.delaunay_beam(scenes.L1M if what.endswith("1m") else scenes.L100K)
        g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
        scenes.build_unstructured_pd(g, mesh)
        g.finalize()
        for _ in range(6):
            g.tick_async(1)
            g.synchronize()
        return g
    raise SystemExit("unknown workload " + what)


def main():
    whats = [a for a in sys.argv[1:] if "=" not in a]
    for kv in sys.argv[1:]:
        if "=" in kv:
            name, _, value = kv.partition("=")
            capi.set_tuning(name, value)
    for what in whats:
        t0 = time.perf_counter()
        g = make(what)
        build_s = time.perf_counter() - t0
        n, nnz = g.count(capi.NODES), g.count(capi.SYSTEM_NNZ)
        ticks = 5 if n > 400000 else 20
        el = bench.timed_ticks(g, ticks, 2, lambda: None)
        B = bench.pd_bytes(g)
        print("%s: n %d nnz %d (%.2f/row) stencils %d  %.1f substeps/s  budget %s  build %.1f s  failed %s" % (
            what, n, nnz, nnz / n, g.count(capi.ROW_STENCILS), ticks / el, g.pcg_health().get("budget"), build_s, g.failed), flush=True)
        for cls in ("pd_spmv", "pd_rhs", "pd_local_tet"):
            cnt, ms, units, ov = g.profile_in_situ(bench.K[cls], 1 if n > 400000 else 2)
            if cnt:
                net = max(ms - cnt * ov, 0.05 * ms)
                per = B[cls]
                print("   in situ %-13s %4d brackets  avg %8.2f us (bracket %.2f, overhead %.2f)  %7.1f GB/s = %.3f by %.1f B/unit" % (
                    cls, cnt, 1e3 * net / cnt, 1e3 * ms / cnt, 1e3 * ov, per * units / (net * 1e-3) / 1e9,
                    per * units / (net * 1e-3) / 1e9 / 8000.0, per), flush=True)
        g.close()


if __name__ == "__main__":
    main()
