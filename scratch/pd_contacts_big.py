"""The bench's pd_contacts scene, a few eager ticks for rocprofv3."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from pies_amd import capi
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10))
g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
for k in range(int(os.environ.get("TICKS", "8"))):
    g.tick_async(1); g.synchronize()
print("contacts", len(g.tri_collisions), g.pcg_stats(), g.failed)
