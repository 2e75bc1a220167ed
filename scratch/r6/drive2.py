import ctypes, numpy as np, sys
exec(open("drive.py").read().split("rng = np.random.default_rng(1)")[0])
rng = np.random.default_rng(1)
def report(name, A):
    exp, S = fixed64(A)
    with np.errstate(all="ignore"): cond = S[:,0]/S[:,2]
    good = cond < 1e3
    for mode in (0,1):
        out, st, sv = run(mode, A)
        err = np.abs(out-exp).max(axis=(1,2))
        g = st[:len(st)//64*64].reshape(-1,64,8)
        e = err[good].max() if good.any() else float("nan")
        if mode == 0:
            print("%-24s old: err(cond<1e3) %.2e all %.2e | rot %.2f sweeps %.2f wave-sweeps %.2f" % (name, e, err.max(), st[:,0].mean(), st[:,1].mean(), g[:,:,1].max(1).mean() if len(g) else 0))
        else:
            print("%-24s new: err(cond<1e3) %.2e all %.2e | analytic %.3f rot %.3f sweeps %.3f | wave: analytic-any %.3f sweeps %.3f sw1 rot any 01 %.3f 02 %.3f 12 %.3f" % ("", e, err.max(), st[:,6].mean(), st[:,0].mean(), st[:,1].mean(), g[:,:,6].any(1).mean(), g[:,:,1].max(1).mean(), g[:,:,2].any(1).mean(), g[:,:,3].any(1).mean(), g[:,:,4].any(1).mean()))
if __name__ == "__main__":
    d = np.load("../../tests/golden/svd_fixed.npz")
    report("golden svd_fixed", d["A"])
    def rot(n):
        q,_ = np.linalg.qr(rng.normal(size=(n,3,3))); return q
    n = 64*2000
    for eps in (0.3, 0.1, 0.01, 1e-5):
        report("R(I+%g N)"%eps, rot(n) @ (np.eye(3) + eps*rng.normal(size=(n,3,3))))
    report("normal", rng.normal(size=(n,3,3)))
    report("identity", np.tile(np.eye(3), (640,1,1)))
    report("zeros", np.zeros((640,3,3)))
    report("double sigma", rot(n) @ (np.array([1.3,1.3,0.7])[None,:,None]*rot(n).transpose(0,2,1)))
    report("double sigma low", rot(n) @ (np.array([1.3,0.7,0.7])[None,:,None]*rot(n).transpose(0,2,1)))
    report("triple", 1.3*rot(n))
    report("flat 1e-4", rot(n) @ (np.array([1.3,0.9,1e-4])[None,:,None]*rot(n).transpose(0,2,1)))
    report("needle", rot(n) @ (np.array([1.3,1e-4,1e-4])[None,:,None]*rot(n).transpose(0,2,1)))
    report("rank2 exact col0", rng.normal(size=(n,3,3))*np.array([0,1,1])[None,None,:])
    report("rank2 generic", rot(n) @ (np.array([1.3,0.9,0.0])[None,:,None]*rot(n).transpose(0,2,1)))
    report("tiny 1e-12", 1e-12*rng.normal(size=(n,3,3)))
    report("huge 1e6", 1e6*rng.normal(size=(n,3,3)))
