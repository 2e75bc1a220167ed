#!/usr/bin/env python3
"""Finds the 12-colour tables scene.cpp proposes for the tetrahedra of one lattice layer (schedule LAYERED).

A layer of createTetBox cells perpendicular to axis `a` holds six tetrahedra per cell; 12 of them meet at a node, so
12 colours is a lower bound.  A colouring that is periodic with period 2 in both in-layer directions is searched on
the 2x2 torus of cells (tabu search on the conflict graph; wrap-around only adds conflicts, so a torus colouring is
valid on the open lattice).  The tables are proposals only: layer_plan.cpp verifies them constraint by constraint.
"""
import random

# the six tetrahedra of a cell in the order scene.cpp pushes them (PrimitiveUtilities.cpp:401-514)
Q = [((0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)), ((0, 0, 0), (0, 1, 0), (0, 1, 1), (1, 1, 1)),
     ((0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1)), ((0, 0, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1)),
     ((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1)), ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1))]


def tabucol(adj, k, iters, seed):
    n = len(adj)
    rnd = random.Random(seed)
    col = [rnd.randrange(k) for _ in range(n)]
    gamma = [[0] * k for _ in range(n)]
    for v in range(n):
        for u in adj[v]:
            gamma[v][col[u]] += 1
    conf = sum(gamma[v][col[v]] for v in range(n)) // 2
    tabu, best = {}, conf
    for it in range(iters):
        if conf == 0:
            return col
        cand = [v for v in range(n) if gamma[v][col[v]] > 0]
        bestd, moves = None, []
        for v in cand:
            for c in range(k):
                if c == col[v]:
                    continue
                d = gamma[v][c] - gamma[v][col[v]]
                if tabu.get((v, c), -1) > it and conf + d >= best:
                    continue
                if bestd is None or d < bestd:
                    bestd, moves = d, [(v, c)]
                elif d == bestd:
                    moves.append((v, c))
        if not moves:
            continue
        v, c = rnd.choice(moves)
        cv = col[v]
        for u in adj[v]:
            gamma[u][cv] -= 1
            gamma[u][c] += 1
        col[v] = c
        conf += bestd
        tabu[(v, cv)] = it + rnd.randrange(10) + int(0.6 * len(cand))
        best = min(best, conf)
    return None


def table(axis):
    inlayer = [a for a in range(3) if a != axis]
    tets = []
    for u in range(2):
        for v in range(2):
            for rel in Q:
                nodes = set()
                for p in rel:
                    q = list(p)
                    q[inlayer[0]] = (q[inlayer[0]] + u) % 2
                    q[inlayer[1]] = (q[inlayer[1]] + v) % 2
                    nodes.add(tuple(q))
                tets.append(nodes)
    adj = [set() for _ in tets]
    for a in range(len(tets)):
        for b in range(a + 1, len(tets)):
            if tets[a] & tets[b]:
                adj[a].add(b)
                adj[b].add(a)
    for seed in range(50):
        col = tabucol(adj, 12, 20000, seed)
        if col:
            return [[col[(u * 2 + v) * 6 + e] for e in range(6)] for u in range(2) for v in range(2)]
    raise SystemExit("no 12-colouring found for axis %d" % axis)


if __name__ == "__main__":
    print("// kLayerTetColour[axis][2 * (u & 1) + (v & 1)][e]: u, v = cell indices along the two in-layer axes (ascending)")
    print("static const uint8_t kLayerTetColour[3][4][6] = {")
    for axis in range(3):
        t = table(axis)
        print("    {" + ", ".join("{" + ", ".join(str(c) for c in row) + "}" for row in t) + "},")
    print("};")
