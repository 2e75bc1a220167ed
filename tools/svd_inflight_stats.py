import os, sys, numpy as np, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "benchlib")]
import oracle_api as ora, scenes
L = ora.lib()
L.ora_svd_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
o = ora.OracleSolver(scenes.pbd_options(ora, 20))
scenes.build_beam(o, (20,20,60)); scenes.perturb(o, 1234, 0.05); o.set_flag(1, 0)
if len(sys.argv) > 1: o.set_flag(5, int(sys.argv[1]))
st = (ctypes.c_uint64*8)()
for t in range(4):
    o.tick(1 if t < 3 else 10)
    L.ora_svd_stats(st, 1)
    a = np.array(st[:], dtype=np.float64)
    print("tick batch %d: calls %d closed-form %.3f rotations/call %.3f sweeps/call %.3f hist(1,2,3,>=4) %s" % (t, a[0], a[1]/a[0], a[2]/a[0], a[3]/a[0], (a[4:]/a[0]).round(4)))
