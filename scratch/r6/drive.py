import ctypes, numpy as np, sys
L = ctypes.CDLL("./proto.so")
pf = ctypes.POINTER(ctypes.c_float); pi = ctypes.POINTER(ctypes.c_int)
L.fixed_batch.argtypes = [ctypes.c_int, ctypes.c_int, pf, ctypes.c_float, ctypes.c_float, pf, pi, pf]
L.set_tol.argtypes = [ctypes.c_float]
def run(mode, A, lo=0.8, hi=1.0):
    A = np.ascontiguousarray(A, dtype=np.float32); n = len(A)
    out = np.zeros((n,3,3), np.float32); st = np.zeros((n,8), np.int32); sv = np.zeros((n,3), np.float32)
    L.fixed_batch(mode, n, A.ctypes.data_as(pf), lo, hi, out.ctypes.data_as(pf), st.ctypes.data_as(pi), sv.ctypes.data_as(pf))
    return out, st, sv
def fixed64(F, lo=0.8, hi=1.0):
    F = F.astype(np.float64)
    U,S,Vt = np.linalg.svd(F)
    Sn = np.clip(S, lo, hi)
    neg = np.linalg.det(F) < 0
    Sn[neg,2] *= -1
    return (U*Sn[:,None,:])@Vt, S
rng = np.random.default_rng(1)
def report(name, A):
    exp, S = fixed64(A)
    cond = S[:,0]/S[:,2]
    for mode in (0,1):
        out, st, sv = run(mode, A)
        err = np.abs(out-exp).max(axis=(1,2))
        good = cond < 1e3
        print("%-28s mode %d: err max(cond<1e3) %.2e  p99 %.2e max(all) %.2e | rotations mean %.2f max %d sweeps mean %.2f" % (name, mode, err[good].max(), np.quantile(err[good],0.99), err.max(), st[:,0].mean(), st[:,0].max(), st[:,1].mean()), end="")
        if mode == 1:
            print(" | first01 %.3f then02 %.4f 12 %.4f re01 %.4f" % (st[:,2].mean(), st[:,3].mean(), st[:,4].mean(), st[:,5].mean()))
            # wave-level: groups of 64
            g = st[:len(st)//64*64].reshape(-1,64,8)
            print("     wave-level: any02 %.3f any12 %.3f anyre01 %.3f  extra sweeps mean %.3f" % (g[:,:,3].any(1).mean(), g[:,:,4].any(1).mean(), g[:,:,5].any(1).mean(), (g[:,:,1].max(1)-1).mean()))
        else:
            g = st[:len(st)//64*64].reshape(-1,64,8)
            print("\n     wave-level sweeps (max over 64): %.2f" % g[:,:,1].max(1).mean())
d = np.load("../../tests/golden/svd_fixed.npz")
report("golden svd_fixed", d["A"])
def rot(n):
    q,_ = np.linalg.qr(rng.normal(size=(n,3,3))); return q
n = 64*2000
for eps in (0.3, 0.1, 0.03, 0.01, 1e-3, 1e-5):
    A = rot(n) @ (np.eye(3) + eps*rng.normal(size=(n,3,3)))
    report("R(I+%g N)"%eps, A)
report("normal", rng.normal(size=(n,3,3)))
report("identity-ish exact", np.tile(np.eye(3), (640,1,1)))
report("zeros", np.zeros((640,3,3)))
A = rot(n) @ (np.array([1.3,1.3,0.7])[None,:,None]*rot(n).transpose(0,2,1)); report("double sigma", A)
A = rot(n) @ (np.array([1.3,0.7,0.7])[None,:,None]*rot(n).transpose(0,2,1)); report("double sigma low", A)
A = rot(n) @ (np.array([1.3,0.9,1e-4])[None,:,None]*rot(n).transpose(0,2,1)); report("flat", A)
A = rot(n) @ (np.array([1.3,1e-4,1e-4])[None,:,None]*rot(n).transpose(0,2,1)); report("needle", A)
