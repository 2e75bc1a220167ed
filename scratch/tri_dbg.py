import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from pies_amd import capi as pies
import oracle_api as oracle
from test_pd_parity_gpu import pd_options, tol_for
g = pies.Solver(pd_options(pies, 3)); g.set_pcg(3e-7, 256); o = oracle.OracleSolver(pd_options(oracle, 3))
for s in (g, o):
    s.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
    s.create_tet_box(12, 2, 18, translation=(0.37, 1.05, 0.41), w=1.0)
    v = s.velocities; v[14*2*20:, 1] = -1.5; s.set_velocities(v); s.set_prev_positions(s.positions)
tol = tol_for(o.positions)
for t in range(3):
    g.set_positions(o.positions); g.set_prev_positions(o.prev_positions); g.set_velocities(o.velocities)
    g.tick(); o.tick()
    d = np.abs(g.positions - o.positions)
    print(t, "contacts", len(g.tri_collisions), len(o.tri_collisions), "equal", np.array_equal(g.tri_collisions, o.tri_collisions), "max dpos %.3g tol %.3g at node %d" % (d.max(), tol, d.max(axis=1).argmax()), "pcg", g.pcg_stats(), "dvel %.3g" % np.abs(g.velocities-o.velocities).max())
