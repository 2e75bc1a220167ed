import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pies_amd import capi
import scenes
s = capi.Solver(scenes.pbd_options(capi, 4), device=capi.DEVICE_NONE if hasattr(capi,'DEVICE_NONE') else -1)
scenes.build_beam(s, (20,20,12))
tet = s.ids(capi.TET).reshape(-1,4); dist = s.ids(capi.DISTANCE).reshape(-1,2)
N = s.count(9)
print(N, tet.shape, dist.shape)
z = np.arange(N) % 12   # level = z index
def greedy(ops, wmask, order=None):
    usedW = {}; usedR = {}
    col = np.zeros(len(ops), int)
    for c in (order if order is not None else range(len(ops))):
        forb = set()
        for k,n in enumerate(ops[c]):
            forb |= usedW.get(n,set())
            if wmask>>k & 1: forb |= usedR.get(n,set())
        x = 0
        while x in forb: x += 1
        col[c] = x
        for k,n in enumerate(ops[c]):
            (usedW if wmask>>k&1 else usedR).setdefault(n,set()).add(x)
    return col
for name, ops, wm in (("tet", tet, 15), ("dist", dist, 1)):
    lv = z[ops].min(axis=1); assert (z[ops].max(axis=1) - lv <= 1).all()
    for L in (0, 5, 10):
        sel = ops[lv == L]
        col = greedy(sel, wm)
        best = col.max()+1
        # iterated greedy
        for r in range(20):
            classes = [np.where(col==k)[0] for k in range(col.max()+1)]
            if r%3==0: classes = classes[::-1]
            elif r%3==1: classes.sort(key=lambda a:-len(a))
            else: classes.sort(key=lambda a:len(a))
            order = np.concatenate(classes)
            col = greedy(sel, wm, order); best = min(best, col.max()+1)
        print(name, "layer", L, "count", len(sel), "colours", best, "sizes", np.bincount(col))
