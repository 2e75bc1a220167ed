import sys, numpy as np, ctypes, time
sys.path.insert(0, "benchlib"); sys.path.insert(0, ".")
import oracle_api as ora, scenes
src = open("scratch/r6/drive.py").read().split("rng = np.random.default_rng(1)")[0].replace('"./proto.so"', '"scratch/r6/proto.so"')
exec(src)
dims = tuple(int(a) for a in sys.argv[1:4]); nt = int(sys.argv[4])
o = ora.OracleSolver(scenes.pbd_options(ora, 20))
scenes.build_beam(o, dims)
scenes.perturb(o, 1234, 0.05)
o.set_flag(1, 0)
ids = o.ids(ora.TET).reshape(-1,4)
qinv = o.rest(ora.TET).reshape(-1,3,3)   # [col][row]
def Fs():
    x = o.positions.astype(np.float32)
    P = np.stack([x[ids[:,1]]-x[ids[:,0]], x[ids[:,2]]-x[ids[:,0]], x[ids[:,3]]-x[ids[:,0]]], axis=1)
    return np.einsum("tkr,tck->tcr", P, qinv).astype(np.float32)
t0 = time.time()
for tick in range(0, nt+1):
    if tick in (0,1,2,3,5,10,20,25,30,35,40):
        A = Fs()
        sv = np.linalg.svd(A.astype(np.float64), compute_uv=False)
        x = o.positions
        print("tick", tick, "t=%.0fs bbox %s..%s sigma q01 %.3g q50 %.3g q99 %.3g mean|s-1| %.3f" % (time.time()-t0, x.min(0).round(1), x.max(0).round(1), np.quantile(sv,0.01), np.quantile(sv,0.5), np.quantile(sv,0.99), np.abs(sv-1).mean()), flush=True)
        for mode in (0,1):
            out, st, s = run(mode, A)
            g = st[:len(st)//64*64].reshape(-1,64,8)
            if mode == 0: print("   old: rotations mean %.2f, sweeps mean %.2f, wave-level sweeps %.2f rot %.2f" % (st[:,0].mean(), st[:,1].mean(), g[:,:,1].max(1).mean(), 0), flush=True)
            else: print("   new: rotations mean %.3f; lane 02 %.4f 12 %.4f re01 %.4f; wave any02 %.3f any12 %.3f anyre01 %.3f extra sweeps %.3f" % (st[:,0].mean(), st[:,3].mean(), st[:,4].mean(), st[:,5].mean(), g[:,:,3].any(1).mean(), g[:,:,4].any(1).mean(), g[:,:,5].any(1).mean(), (g[:,:,1].max(1)-1).mean()), flush=True)
    o.tick(1)
