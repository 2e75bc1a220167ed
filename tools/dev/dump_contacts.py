import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from pies_amd import capi
import bench
g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
g.finalize()
for _ in range(20):
    g.tick_async(1); g.synchronize()
np.save("gpurun_out/r2i/pdc_contacts.npy", g.tri_collisions)
g.close()
g = bench.contact_scene(capi, 0)
g.finalize()
for _ in range(2):
    g.tick_async(1); g.synchronize()
np.save("gpurun_out/r2i/c5_contacts.npy", g.tri_collisions)
