import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import bench, scenes
from pies_amd import capi
mesh = scenes.delaunay_beam(scenes.L100K)
def run(mesh, tag):
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
    scenes.build_unstructured_pd(g, mesh)
    g.finalize()
    for _ in range(20):
        g.tick_async(1); g.synchronize()
    el = bench.timed_ticks(g, 20, 2, lambda: None)
    print(tag, "%.1f substeps/s" % (20 / el), "nnz", g.count(capi.SYSTEM_NNZ), "window entries", g.count(capi.PD_WINDOW_ENTRIES), "halo/row %.2f" % (g.count(capi.PD_WINDOW_HALO) / g.count(capi.NODES)), "tiles", g.count(capi.PD_TILES), "records/node %.2f" % (g.count(capi.PD_TILE_RECORDS) / g.count(capi.NODES)))
    for name in ("pd_spmv", "pd_rhs", "pd_local_tet"):
        n, ms, units, ov = g.profile_in_situ(bench.K[name], 2)
        if n: print("   in situ %-12s: avg %.2f us" % (name, 1e3 * ms / n - 1e3 * ov))
    g.close()
run(mesh, "lattice order (z fastest)")
pos, tets, edges = mesh
# blocked order: 5 x 5 x 10 bricks
W, H, D = scenes.L100K
ijk = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3)
for bx, by, bz in ((5, 5, 10), (4, 4, 16), (10, 10, 5)):
    key = ((ijk[:, 2] // bz) * 1000 + (ijk[:, 0] // bx) * 30 + (ijk[:, 1] // by)) * 100000 + ((ijk[:, 0] % bx) * by + (ijk[:, 1] % by)) * bz + (ijk[:, 2] % bz)
    order = np.argsort(key, kind="stable")       # new -> old
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    m2 = (pos[order], inv[tets].astype(np.uint32), np.sort(inv[edges], axis=1).astype(np.uint32))
    run(m2, "bricks %dx%dx%d" % (bx, by, bz))
rng = np.random.default_rng(1)
order = rng.permutation(len(pos)); inv = np.empty_like(order); inv[order] = np.arange(len(order))
run((pos[order], inv[tets].astype(np.uint32), np.sort(inv[edges], axis=1).astype(np.uint32)), "random order")
