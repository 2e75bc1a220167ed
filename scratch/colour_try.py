import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from pies_amd import capi
import scenes
dims = scenes.L100K if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split('x'))
s = capi.Solver(scenes.pbd_options(capi, 20), device=-1)
scenes.build_beam(s, dims)
s.set_schedule(capi.SCHEDULE_COLOURED)
t=time.time(); s.finalize(); print('finalize %.2fs'%(time.time()-t))
for ty,name in ((capi.DISTANCE,'dist'),(capi.TET,'tet')):
    b = s.batches(ty); print(name, len(b)-1, np.diff(b).tolist())
