#!/usr/bin/env python3
"""Randomised soak of the node-node pass in the pair order (development aid): loose particles with random sizes, radii, grid
spacings (cell ranges of 2 to 4 cells per axis), jitter and iteration counts; two ticks each against the oracle's rule 2, exact
equality, and the device run repeated (bit-identical).  usage: soak_collisions.py [scenes] [seed] [pairs|turns]
(turns: the reference's order executed by dependency levels of turns - PIES_REFERENCE_TURNS=1 whatever the scene's size - against
the oracle's plain loop, rule 0)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import oracle_api as ora  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402


def main(nscenes, seed, order="pairs", max_dim=34):
    """order: pairs (the device's pair order against the oracle's rule 2), turns (the reference's order by dependency levels of
    turns against the oracle's plain loop), groups (the group order against the oracle's rule 1)"""
    turns = order == "turns"
    capi.set_tuning("PIES_REFERENCE_TURNS", "1" if turns else None)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for sc in range(nscenes):
        dims = tuple(int(v) for v in rng.integers(4, max_dim, 3))
        spacing = float(rng.choice([0.8, 0.9, 1.0]))
        jitter = float(rng.choice([0.02, 0.05, 0.15]))
        grid = float(rng.choice([2.0, 2.0, 1.0, 0.6]))
        iters = int(rng.integers(1, 5))
        p = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1).reshape(-1, 3) * spacing
        p = (p + rng.uniform(-jitter, jitter, p.shape) + [0, 0.5, 0]).astype(np.float32)
        v = rng.uniform(-1, 1, p.shape).astype(np.float32)
        r = np.full(len(p), 0.5, np.float32)
        if rng.random() < 0.4:
            r[rng.random(len(p)) < 0.15] = 0.35
        state = (dims, spacing, jitter, grid, iters)
        res = []
        for which in ("oracle", "device", "device"):
            mod = ora if which == "oracle" else capi
            s = (mod.OracleSolver if which == "oracle" else mod.Solver)(scenes.pbd_options(mod, iters, gridSpacing=grid))
            s.add_nodes_raw(p, vel=v, radius=r, invMass=np.ones(len(p), np.float32))
            if which == "oracle":
                # (the group order needs cell ranges of at most 2 cells per axis: with a finer grid the device runs the reference's loop)
                s.set_flag(ora.FLAG_COLLISION_RULE, 0 if turns else (1 if grid >= 2.0 else 0) if order == "groups" else 2)
            else:
                s.set_flag(capi.FLAG_COLLISION_ORDER, capi.COLLISION_ORDER_REFERENCE if turns else
                           capi.COLLISION_ORDER_GROUPS if order == "groups" else capi.COLLISION_ORDER_PAIRS)
            s.tick(2)
            res.append((s.positions.copy(), s.velocities.copy(), s.collision_pairs, s.failed))
            if which != "oracle":
                h = dict(s.collision_health(), fallbacks=s.collision_fallbacks)
                s.close()
        assert not res[1][3], ("failed", state)
        for k in range(2):
            assert np.array_equal(res[0][k], res[1][k]), ("device vs oracle", k, state, float(np.abs(res[0][k] - res[1][k]).max()))
            assert np.array_equal(res[1][k], res[2][k]), ("run to run", k, state)
        assert res[0][2] == res[1][2] == res[2][2], ("pairs", state, res[0][2], res[1][2])
        print("scene %d ok: %s, %d particles, %d resolved pairs, health %s, %.0f s" % (sc, state, len(p), res[1][2], h, time.time() - t0), flush=True)
    capi.set_tuning("PIES_REFERENCE_TURNS", None)
    print("done: %d scenes" % nscenes)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 5, sys.argv[3] if len(sys.argv) > 3 else "pairs")
