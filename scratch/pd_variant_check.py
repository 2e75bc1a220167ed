"""The two graph variants of the PD global step (contact rows inline / in a pass of their own) on a scene with thousands
of binding contacts: positions after a few ticks must agree to PD tolerance."""
import sys, os, subprocess
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
if len(sys.argv) > 1:
    from pies_amd import capi
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=int(os.environ.get("ITERS", "10"))))
    g.create_tet_box(10,10,60, translation=(0,0.04,0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(10,10,15, translation=(0.3, 0.04 + 9 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
    g.finalize()
    for k in range(int(os.environ.get("TICKS", "8"))): g.tick()
    print("contacts", len(g.tri_collisions), "pcg", g.pcg_stats(), file=sys.stderr)
    np.save(sys.argv[1], g.positions)
else:
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, "/tmp/pd_variant_%s.npy" % v], env=dict(os.environ, PIES_TRI_FAST_ROWS=v))
    a, b = np.load("/tmp/pd_variant_0.npy"), np.load("/tmp/pd_variant_1.npy")
    print("max |dpos| between the variants after TICKS ticks: %.3g (bbox diagonal %.1f)" % (np.abs(a - b).max(), np.linalg.norm(a.max(0) - a.min(0))))
