"""Prototype: dependency-level depth of the pair-ordered node-node pass (rule 2) on config 4's particles."""
import numpy as np, sys, time
from scipy.spatial import cKDTree
W,H,D = (25,50,50) if len(sys.argv)<2 else tuple(int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(1234)
p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
p = (p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]).astype(np.float32)
n = len(p); r = 0.5
for slack in (0.15, 0.225, 0.3):
    cut = 2*r + 2*slack
    t = cKDTree(p)
    pairs = t.query_pairs(cut, output_type='ndarray')
    i, j = pairs[:,0].astype(np.uint64), pairs[:,1].astype(np.uint64)
    for keyname in ("hash", "index"):
        if keyname == "hash":
            k = (i * np.uint64(0x9E3779B97F4A7C15) ^ (j * np.uint64(0xC2B2AE3D27D4EB4F)))
            k = (k ^ (k >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
            k ^= k >> np.uint64(32)
        else:
            k = i * np.uint64(n) + j
        order = np.argsort(k, kind='stable')
        ii, jj = pairs[order,0], pairs[order,1]
        lvl_node = np.zeros(n, np.int32)
        # sequential level computation (python loop is slow; vectorise by chunks impossible) -> use simple loop in numba-free way
        lv = np.empty(len(ii), np.int32)
        t0=time.time()
        ln = lvl_node.tolist(); a=ii.tolist(); b=jj.tolist(); out=[0]*len(a)
        for e in range(len(a)):
            l = max(ln[a[e]], ln[b[e]]) + 1
            ln[a[e]] = l; ln[b[e]] = l; out[e]=l
        lv = np.array(out)
        depth = lv.max()
        hist = np.bincount(lv)[1:]
        deg = np.bincount(np.concatenate([ii,jj]), minlength=n)
        print(f"slack {slack} cut {cut:.2f} key {keyname}: pairs {len(ii)} deg avg {deg.mean():.1f} max {deg.max()} depth {depth}; "
              f"edges/level first10 {hist[:10].tolist()} ; levels with <1% of edges: {(hist < 0.01*len(ii)/depth).sum()} ; cum frac by level:",
              [round(float(hist[:q].sum()/len(ii)),3) for q in (10,20,30,40,50,60,80)])
