"""Rounds of the pair order with and without chaining (a lane that finished a pair goes on with a node's next pair when that
pair's other node has been waiting since an earlier round).  62.5k jittered particles, pairs within a cut, keys = class+parity+hash."""
import numpy as np, sys
from scipy.spatial import cKDTree
W,H,D = (25,50,50)
rng = np.random.default_rng(1234)
p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
jit = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
p = (p + rng.uniform(-jit, jit, p.shape)).astype(np.float32)
n = len(p)
pairs = cKDTree(p).query_pairs(cut, output_type='ndarray')
i, j = pairs[:,0].astype(np.uint64), pairs[:,1].astype(np.uint64)
h = (i * np.uint64(0x9E3779B97F4A7C15) ^ (j * np.uint64(0xC2B2AE3D27D4EB4F)))
h = (h ^ (h >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
d = (p[pairs[:,1]] - p[pairs[:,0]]).astype(np.float32)
ad = np.abs(d); m = ad.max(1, keepdims=True)
q = np.where(ad > np.float32(0.41421356) * m, np.sign(d), 0).astype(np.int32)
first = np.argmax(q != 0, axis=1); sgn = q[np.arange(len(q)), first]; q = q * sgn[:, None]
cls = (q[:,0] + 1) * 9 + (q[:,1] + 1) * 3 + (q[:,2] + 1)
qq = (q * q).sum(1).astype(np.float32)
ui = (p[pairs[:,0]] * q).sum(1) / qq; uj = (p[pairs[:,1]] * q).sum(1) / qq
L = np.abs(uj - ui); umin = np.minimum(ui, uj)
par = (np.floor(umin / np.maximum(L, 1e-6)).astype(np.int64) & 1).astype(np.uint64)
key = ((cls.astype(np.uint64) * np.uint64(2) + par) << np.uint64(58)) | (h >> np.uint64(6))
order = np.argsort(key, kind='stable')
lists = [[] for _ in range(n)]
for e in order:
    a, b = int(pairs[e,0]), int(pairs[e,1])
    lists[a].append(b); lists[b].append(a)
deg = np.array([len(l) for l in lists])
print("nodes", n, "pairs", len(pairs), "deg mean %.1f max %d" % (deg.mean(), deg.max()))

def run(K):
    cur = [0]*n; stamp = [0]*n    # stamp: round in which the node reached its current entry
    frontier = list(range(n)); rnd = 1; longest = 0; total_chain = 0
    while frontier:
        nxt = []
        def entry(u): return lists[u][cur[u]] if cur[u] < len(lists[u]) else -1
        def move(u):
            cur[u] += 1; stamp[u] = rnd
            return cur[u] < len(lists[u])
        for x in frontier:
            if cur[x] >= len(lists[x]) or stamp[x] != rnd - 1: continue
            y = entry(x)
            if cur[y] >= len(lists[y]) or entry(y) != x or stamp[y] == rnd: continue
            if stamp[y] == rnd - 1 and y < x: continue
            heads = []
            if move(x): heads.append(x)
            if move(y): heads.append(y)
            work = 1
            for u in heads:
                alive = True
                for it in range(K):
                    z = entry(u)
                    if cur[z] < len(lists[z]) and entry(z) == u and stamp[z] != rnd:
                        hu = move(u); hz = move(z); work += 1
                        if hz: nxt.append(z)
                        if not hu: alive = False; break
                    else:
                        break
                if alive: nxt.append(u)
            longest = max(longest, work); total_chain += work - 1
        frontier = nxt; rnd += 1
    assert all(cur[u] == len(lists[u]) for u in range(n))
    return rnd - 1, longest, total_chain
for K in (0, 1, 2, 3, 8):
    r, lw, tc = run(K)
    print("chain limit %d per head: rounds %d, most pairs in one lane %d, chained pairs %d" % (K, r, lw, tc))
