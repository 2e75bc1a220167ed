"""Prototype: level depth of the pair order with geometric class keys vs hashed keys (config 4 particles, relaxed a bit)."""
import numpy as np, sys
from scipy.spatial import cKDTree
W,H,D = (25,50,50)
rng = np.random.default_rng(1234)
p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
jit = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
p = (p + rng.uniform(-jit, jit, p.shape) + [0, 0.5, 0]).astype(np.float32)
n = len(p)
def depth_of(ii, jj, k):
    order = np.argsort(k, kind='stable')
    a = ii[order].tolist(); b = jj[order].tolist(); ln=[0]*n; top=0
    for e in range(len(a)):
        l = max(ln[a[e]], ln[b[e]]) + 1; ln[a[e]] = l; ln[b[e]] = l
        if l>top: top=l
    return top
for cut in (1.3, 1.45, 1.6):
    pairs = cKDTree(p).query_pairs(cut, output_type='ndarray')
    i, j = pairs[:,0].astype(np.uint64), pairs[:,1].astype(np.uint64)
    h = (i * np.uint64(0x9E3779B97F4A7C15) ^ (j * np.uint64(0xC2B2AE3D27D4EB4F)))
    h = (h ^ (h >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
    d = (p[pairs[:,1]] - p[pairs[:,0]]).astype(np.float32)
    ad = np.abs(d); m = ad.max(1, keepdims=True)
    q = np.where(ad > np.float32(0.41421356) * m, np.sign(d), 0).astype(np.int32)
    # canonical orientation: first nonzero component positive
    first = np.argmax(q != 0, axis=1)
    sgn = q[np.arange(len(q)), first]
    q = q * sgn[:, None]
    cls = (q[:,0] + 1) * 9 + (q[:,1] + 1) * 3 + (q[:,2] + 1)  # 0..26, only 13 used
    qq = (q * q).sum(1).astype(np.float32)
    ui = (p[pairs[:,0]] * q).sum(1) / qq; uj = (p[pairs[:,1]] * q).sum(1) / qq
    L = np.abs(uj - ui); umin = np.minimum(ui, uj)
    par = (np.floor(umin / np.maximum(L, 1e-6)).astype(np.int64) & 1).astype(np.uint64)
    shell = np.minimum((np.linalg.norm(d, axis=1) / 0.7).astype(np.int64), 3).astype(np.uint64)
    k1 = ((cls.astype(np.uint64) * np.uint64(2) + par) << np.uint64(58)) | (h >> np.uint64(6))
    k2 = ((shell * np.uint64(54) + cls.astype(np.uint64) * np.uint64(2) + par) << np.uint64(56)) | (h >> np.uint64(8))
    deg = np.bincount(np.concatenate([pairs[:,0], pairs[:,1]]), minlength=n)
    print(f"jitter {jit} cut {cut}: pairs {len(pairs)} deg mean {deg.mean():.1f} max {deg.max()}  depth hash {depth_of(pairs[:,0], pairs[:,1], h)}  class+parity {depth_of(pairs[:,0], pairs[:,1], k1)}  shell+class+parity {depth_of(pairs[:,0], pairs[:,1], k2)}  classes used {len(np.unique(cls))}")
