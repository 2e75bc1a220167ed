import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import scenes, oracle_api as oracle
from pies_amd import capi as pies
from test_pd_parity_gpu import pd_options, build_pd_beam
def make(mod, fp64=False):
    s = mod.Solver(pd_options(mod, 10)) if mod is pies else mod.OracleSolver(pd_options(mod, 10))
    build_pd_beam(s, scenes.L100K, translation=(0.0, 2.0, 0.0))
    scenes.perturb(s, 21, 0.03); s.set_prev_positions(s.positions)
    if fp64: s.set_flag(oracle.FLAG_PD_SOLVE_FP64, 1)
    return s
o32, o64 = make(oracle), make(oracle, True)
O64, REF = [], []
for t in range(2):
    o32.tick(1); o64.tick(1)
    O64.append(o64.positions.copy()); REF.append(float(np.abs(o32.positions - o64.positions).max()))
P32 = []
g = make(pies); g.finalize()
for t in range(2):
    g.tick(1)
    d = g.positions - O64[t]
    i, c = np.unravel_index(np.abs(d).argmax(), d.shape)
    print("tick", t, "max dev node", i, "comp", c, "pos", g.positions[i], "dev", d[i], "rms per comp", np.sqrt((d ** 2).mean(0)))
    idx = np.argsort(-np.abs(d).max(1))[:8]
    print("  worst nodes", idx, np.abs(d).max(1)[idx])
    z = g.positions[:, 2]
    for lo in range(0, 250, 50):
        m = (z >= lo) & (z < lo + 50)
        print("   z in [%d,%d): max %.3g rms %.3g  mean signed %s" % (lo, lo + 50, np.abs(d[m]).max(), np.sqrt((d[m] ** 2).mean()), d[m].mean(0)))
    v = g.velocities
    print("  |v| max", np.abs(v).max(), " y min", g.positions[:,1].min())
g.close()
