"""The PD global step is exact in the reference (a sparse Cholesky factorisation per substep, Solver.cpp:258-262, 356);
the device's is a CG with a captured iteration budget that follows what recent solves needed.  When new contacts stiffen
the system from one substep to the next, a solve can end above the tolerance: pies_tick then puts the substep's input
back and runs it again with a larger budget, so every substep it returns met the tolerance; pies_tick_async cannot do
that and counts the short solves instead (pies_get_pcg_health)."""
import numpy as np
import pytest

import scenes
from test_pd_parity_gpu import pd_options, tol_for

pytestmark = pytest.mark.gpu
TOL = 3e-7


def plates(s):
    s.create_tet_box(14, 2, 20, translation=(0, 0.02, 0), w=1.0)
    s.create_tet_box(12, 2, 18, translation=(0.37, 1.35, 0.41), w=1.0)   # 0.3 above the lower plate, falling
    v = s.velocities
    v[14 * 2 * 20:, 1] = -3.0
    s.set_velocities(v)
    s.set_prev_positions(s.positions)


def test_contact_onset_with_the_default_budget_meets_the_tolerance(pies, oracle):
    g = pies.Solver(pd_options(pies, 3))         # default pies_set_pcg: 3e-7, ceiling 128, budget starts at 32 and shrinks
    o = oracle.OracleSolver(pd_options(oracle, 3))
    for s in (g, o):
        plates(s)
    worst, contacts, budgets = 0.0, 0, []
    for t in range(30):
        g.tick(); o.tick()
        res, iters, solves = g.pcg_stats()
        worst = max(worst, res)
        contacts = max(contacts, len(g.tri_collisions))
        budgets.append(g.pcg_health()["budget"])
        assert solves == 3
        assert res <= TOL * 1.0001, (t, res, budgets)          # every substep handed to the host met the tolerance
    h = g.pcg_health()
    assert contacts > 1000                                    # the plates did meet
    assert min(budgets) < 32 <= max(budgets)                  # the budget had shrunk in free fall and grew at the onset
    assert h["substeps_retried"] >= 1 and h["short_solves"] == 0, h
    # and the run as a whole stays with the oracle's exact solves (free run, no teacher forcing: contact decisions are
    # discontinuous, so this is a loose gate on the bulk)
    assert np.abs(g.positions.mean(0) - o.positions.mean(0)).max() < 2e-2


def test_asynchronous_ticks_count_what_they_could_not_repair(pies):
    g = pies.Solver(pd_options(pies, 3))
    g.set_pcg(TOL, 4)                      # a ceiling far too low for 5000 contacts of weight 1e4
    plates(g)
    for t in range(30):
        g.tick_async()
    g.synchronize()
    h = g.pcg_health()
    assert h["solves"] == 90 and h["short_solves"] > 0 and h["substeps_retried"] == 0, h
    # the synchronous tick at its ceiling keeps the substep but reports it the same way
    before = h["short_solves"]
    g.tick()
    assert g.pcg_health()["short_solves"] > before


def test_retry_can_be_switched_off(pies):
    g = pies.Solver(pd_options(pies, 3))
    g.set_pcg_retry(False)
    plates(g)
    g.tick(30)
    assert g.pcg_health()["substeps_retried"] == 0
