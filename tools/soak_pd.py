#!/usr/bin/env python3
"""Randomised soak of the PD contact path (development aid): small two-body scenes with random sizes, offsets and speeds;
every tick is started from the oracle's state and compared after it (contact lists entry for entry, positions within the PD
tolerance), the device run is repeated (bit-identical) and repeated once more with the sequential passes through L2
(PIES_TRI_LDS=0, bit-identical).  usage: soak_pd.py [scenes] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import oracle_api as ora  # noqa: E402
from pies_amd import capi  # noqa: E402
from test_pd_parity_gpu import pd_options, tol_for  # noqa: E402


def build(s, rng_state):
    (w1, d1, w2, d2, ox, oz, gap, vy, iters) = rng_state
    s.create_tet_box(w1, 2, d1, translation=(0, 0.02, 0), w=1.0)
    s.create_tet_box(w2, 2, d2, translation=(ox, 1.02 + gap, oz), w=1.0)
    v = s.velocities
    v[w1 * 2 * d1:, 1] = vy
    s.set_velocities(v)
    s.set_prev_positions(s.positions)


def device_run(state, ticks, forced, env):
    for k, v in env.items():
        capi.set_tuning(k, v)
    g = capi.Solver(pd_options(capi, state[-1]))
    g.set_pcg(3e-7, 256)
    build(g, state)
    out = []
    for t in range(ticks):
        g.set_positions(forced[t][0]); g.set_prev_positions(forced[t][1]); g.set_velocities(forced[t][2])
        g.tick()
        out.append((g.positions.copy(), g.velocities.copy(), g.tri_collisions.copy(), g.pcg_health()["short_solves"], g.failed))
    g.close()
    for k in env:
        capi.set_tuning(k, None)
    return out


def main(nscenes, seed, max_w=15, max_d=21):
    rng = np.random.default_rng(seed)
    worst = 0.0
    t0 = time.time()
    for sc in range(nscenes):
        state = (int(rng.integers(4, max_w)), int(rng.integers(4, max_d)), int(rng.integers(3, max_w - 2)), int(rng.integers(3, max_d - 2)),
                 float(rng.uniform(0.0, 1.5)), float(rng.uniform(0.0, 1.5)), float(rng.uniform(0.01, 0.08)), float(rng.uniform(-2.5, -0.5)),
                 int(rng.integers(2, 6)))
        ticks = 5
        o = ora.OracleSolver(pd_options(ora, state[-1]))
        build(o, state)
        forced, expect = [], []
        for t in range(ticks):
            forced.append((o.positions.copy(), o.prev_positions.copy(), o.velocities.copy()))
            o.tick()
            expect.append((o.positions.copy(), o.tri_collisions.copy()))
        if o.failed:
            print("scene", sc, "oracle failed: skipped", flush=True)
            continue
        a = device_run(state, ticks, forced, {})
        b = device_run(state, ticks, forced, {})
        c = device_run(state, ticks, forced, {"PIES_TRI_LDS": "0"})
        d = device_run(state, ticks, forced, {"PIES_PCG_BUDGET": "2", "PIES_TRI_FAST_ROWS": "1"})
        tol = 2.0 * tol_for(expect[0][0])
        most = 0
        for t in range(ticks):
            assert np.array_equal(a[t][2], expect[t][1]), ("contact list", sc, t, state)
            # twice the tests' PD tolerance; one scene in a thousand goes a little beyond it (seed 23, scene 859: 1.13 x, the same
            # to four digits with rel_tol 3e-7, 1e-7 and 3e-8, so it is not the CG: with hundreds of w = 1e4 contacts on a few
            # nodes the oracle's fp32 Cholesky is itself good to ~6e-5, and four Gauss-Seidel stabilisation passes carry that
            # along their chains) - reported, and a hard failure only beyond 3 x
            err = float(np.abs(a[t][0] - expect[t][0]).max())
            worst = max(worst, err / tol)
            if err > tol:
                print("  scene %d tick %d: %.2f x the tolerance" % (sc, t, err / tol), flush=True)
            assert err <= 1.5 * tol, ("positions", sc, t, err, tol, state)
            errd = float(np.abs(d[t][0] - expect[t][0]).max())
            assert errd <= 1.5 * tol and np.array_equal(d[t][2], expect[t][1]), ("two captured iterations", sc, t, errd, tol, state)
            for k in range(3):
                assert np.array_equal(a[t][k], b[t][k]), ("run to run", sc, t, k, state)
                assert np.array_equal(a[t][k], c[t][k]), ("lds vs l2", sc, t, k, state)
            assert a[t][3] == 0 and d[t][3] == 0 and not a[t][4], ("health", sc, t, state)
            most = max(most, len(expect[t][1]))
        print("scene %d ok: %s, most contacts %d, worst error / tolerance so far %.2f, %.0f s" % (sc, state, most, worst, time.time() - t0), flush=True)
    print("done: %d scenes" % nscenes)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
