#!/usr/bin/env python3
"""Randomised soak of schedule LAYERED's candidate plans (layer_plan.cpp) against the oracle replaying the exported order:
unstructured (Delaunay) beams and lattices of random sizes, every candidate forced in turn, positions bit for bit after two
ticks.  python tools/soak_layered.py [scenes] [first seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle_api as oracle  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402


def main(n, first):
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        dims = (int(rng.integers(4, 9)), int(rng.integers(4, 8)), int(rng.integers(12, 48)))
        lattice = bool(rng.integers(0, 2))
        mesh = None if lattice else scenes.delaunay_beam(dims, seed=seed, jitter=float(rng.uniform(0.1, 0.35)))
        strips = bool(rng.integers(0, 3) == 0)
        for cand in (0, 1, 2):
            capi.set_tuning("PIES_LAYER_PLAN_FORCE", str(cand))
            if strips:  # the strip path on a small scene
                capi.set_tuning("PIES_LAYER_ONE_STRIP_MAX", "60")
                capi.set_tuning("PIES_LAYER_TILE_NODES", "120")
                capi.set_tuning("PIES_LAYER_STRIPS_MIN_NODES", "0")
            g = capi.Solver(scenes.pbd_options(capi, 4), device=0)
            o = oracle.OracleSolver(scenes.pbd_options(oracle, 4))
            for s in (g, o):
                if lattice:
                    scenes.build_beam(s, dims)
                else:
                    scenes.build_unstructured(s, mesh)
                scenes.perturb(s, seed, 0.04)
                s.set_flag(1, 0)
            g.set_schedule(capi.SCHEDULE_LAYERED)
            g.finalize()
            for t in (capi.DISTANCE, capi.TET):
                o.permute(t, g.order(t))
            g.tick(2)
            o.tick(2)
            assert np.array_equal(g.positions, o.positions) and np.array_equal(g.velocities, o.velocities), (seed, dims, lattice, strips, cand)
            layered = g.launch_counts()["layer"]
            g.close()
            for name in ("PIES_LAYER_PLAN_FORCE", "PIES_LAYER_ONE_STRIP_MAX", "PIES_LAYER_TILE_NODES", "PIES_LAYER_STRIPS_MIN_NODES"):
                capi.set_tuning(name, None)
        print("seed %d ok: %s %s%s, layer launches %d" % (seed, dims, "lattice" if lattice else "delaunay", ", strips" if strips else "", layered), flush=True)
    print("done: %d scenes" % n)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
