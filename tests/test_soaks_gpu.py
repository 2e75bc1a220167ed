"""Seeded short runs of the randomised soaks (tools/soak_*.py; the long runs are under profiles/) inside the GPU suite, so that the
driver's own run of `pytest -m gpu` sees them: random triangle soups (point-triangle broad phase + CCD, contact lists entry for
entry), random loose-particle scenes in all three node-node orders (bit for bit against the oracle, run to run), random lattices /
Delaunay beams under every candidate plan of schedule LAYERED (bit for bit against the oracle replaying the exported order) and
random two-body PD contact scenes (contact lists exact, positions within the PD tolerance, run to run, LDS against L2 passes, two
captured CG iterations).  Sized for about a minute in total."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_soak_triangle_soups(pies):
    import oracle_api
    from test_tri_collisions_gpu import run_soup
    contacts = sum(run_soup(pies, oracle_api, seed) for seed in range(600, 610))
    assert contacts > 0


@pytest.mark.parametrize("order", ["pairs", "turns", "groups"])
def test_soak_collision_scenes(pies, order):
    import soak_collisions
    soak_collisions.main(10, 61, order, max_dim=24)


def test_soak_layered_plans(pies):
    import soak_layered
    soak_layered.main(8, 41)


def test_soak_pd_contact_scenes(pies):
    import soak_pd
    soak_pd.main(5, 29, max_w=12, max_d=16)
