"""CPU-only tests of the product's host logic through a host-only handle (PIES_DEVICE_NONE): scene
construction against the oracle's own generators, and the Gauss-Seidel schedules (dependency levels /
colouring).  No compute call is made; tick on such a handle must fail (there is no CPU solver)."""
import ctypes
import os

import numpy as np
import pytest

import oracle_api as O
import scenes
from pies_amd import capi


def host_solver(**kw):
    return capi.Solver(scenes.pbd_options(capi, 4, **kw), device=-1)


def test_host_only_handle_cannot_compute():
    g = host_solver()
    g.addNodes([[0, 1, 0]])
    with pytest.raises(capi.PiesError):
        g.tick()


def test_scene_generators_match_oracle():
    g = host_solver()
    o = O.OracleSolver(scenes.pbd_options(O, 4))
    for s in (g, o):
        scenes.build_beam(s, (4, 5, 6), volume=True, triangles=True)
        s.create_sheet(5, 4, translation=(9, 2, 0), scale=0.5, mass=2.0, w=0.3)
        s.create_bend_sheet(4, 5, translation=(20, 2, 0), scale=1.0, w=0.4)
        s.addNodes([[30, 1, 2], [31, 1, 2]])
    for t in (capi.POSITION, capi.DISTANCE, capi.TET, capi.VOLUME, capi.BEND, capi.TRIANGLES, capi.LINES):
        assert np.array_equal(g.ids(t), o.ids(t)), t
    for t in (capi.DISTANCE, capi.TET, capi.VOLUME, capi.BEND):
        assert np.array_equal(g.rest(t), o.rest(t)), t
    for name in ("positions", "prev_positions", "velocities", "radii", "inv_masses"):
        assert np.array_equal(getattr(g, name), getattr(o, name)), name


def test_shape_and_region_generators_match_oracle():
    g = capi.Solver(capi.Options(solver=capi.PD), device=-1)
    o = O.OracleSolver(O.Options(solver=O.PD))
    region = np.zeros((4, 4), np.float32)
    region[0, 0], region[1, 1], region[2, 2], region[3, 3] = 1.2, 0.8, 2.0, 1.0
    region[3, :3] = (1.0, 1.0, 1.0)
    region[1, 0] = 0.3  # sheared box: exercises the full mat4 inverse
    for s in (g, o):
        s.create_shape_matching_box((0, 0.5, 0), 4, 3, 5, 2.0)
        s.create_shape_matching_sheet(50, 50, translation=(5, 1, 0), scale=0.25, w=3.0)  # the reference's size
        s.add_fixed_regions(region.reshape(16), 7.0)
        s.add_linked_regions(np.stack([region.reshape(16), region.reshape(16) * np.float32(1.5)]), 4.0)
    assert g.count(capi.SHAPE) == o.count(O.SHAPE) and g.count(capi.GOAL) == o.count(O.GOAL) == 1
    for k in range(g.count(capi.SHAPE)):
        assert np.array_equal(g.group_ids(capi.SHAPE, k), o.group_ids(O.SHAPE, k)), k
    assert np.array_equal(g.group_ids(capi.GOAL, 0), o.group_ids(O.GOAL, 0)) and len(g.group_ids(capi.GOAL, 0)) > 0
    assert np.array_equal(g.positions, o.positions) and np.array_equal(g.inv_masses, o.inv_masses)


def _conflict_free(ids, order, offs, writes):
    for b in range(len(offs) - 1):
        sel = ids[order[offs[b]:offs[b + 1]]].reshape(offs[b + 1] - offs[b], -1)
        written = sel[:, writes].ravel()
        assert len(np.unique(written)) == len(written)
        reads = np.setdiff1d(sel.ravel(), written)
        assert len(np.intersect1d(reads, written)) == 0


@pytest.mark.parametrize("schedule", [capi.SCHEDULE_EXACT, capi.SCHEDULE_COLOURED, capi.SCHEDULE_LAYERED])
def test_schedules_are_permutations_of_conflict_free_batches(schedule):
    g = host_solver()
    scenes.build_beam(g, (6, 5, 7))
    g.create_bend_sheet(5, 5, translation=(10, 2, 0))
    g.set_schedule(schedule)
    for t, writes in ((capi.POSITION, [0]), (capi.DISTANCE, [0]), (capi.TET, [0, 1, 2, 3]), (capi.BEND, [0, 1, 2, 3])):
        ids, order, offs = g.ids(t), g.order(t), g.batches(t)
        assert sorted(order.tolist()) == list(range(len(ids)))
        assert offs[0] == 0 and offs[-1] == len(ids) and (np.diff(offs.astype(np.int64)) > 0).all()
        _conflict_free(ids, order, offs, writes)


def test_exact_schedule_keeps_every_conflicting_pair_in_container_order():
    """The property that makes EXACT bit-identical to the sequential sweep."""
    rng = np.random.default_rng(2)
    g = host_solver()
    g.addNodes(rng.uniform(0, 5, (60, 3)))
    tets = np.array([rng.choice(60, 4, replace=False) for _ in range(300)], dtype=np.uint32)
    g.add_tet(tets, 0.1)
    g.set_schedule(capi.SCHEDULE_EXACT)
    order, offs = g.order(capi.TET), g.batches(capi.TET)
    level = np.empty(len(tets), dtype=np.int64)
    for b in range(len(offs) - 1):
        level[order[offs[b]:offs[b + 1]]] = b
    for a in range(len(tets)):
        for b in range(a + 1, len(tets)):
            if len(np.intersect1d(tets[a], tets[b])):
                assert level[a] < level[b]
    for b in range(len(offs) - 1):  # stable inside a level
        assert (np.diff(order[offs[b]:offs[b + 1]].astype(np.int64)) > 0).all()


def test_colour_counts_on_the_reference_lattices():
    g = host_solver()
    scenes.build_beam(g, scenes.L1K)
    g.set_schedule(capi.SCHEDULE_COLOURED)
    nd, nt = len(g.batches(capi.DISTANCE)) - 1, len(g.batches(capi.TET)) - 1
    assert 8 <= nd <= 16 and 24 <= nt <= 32, (nd, nt)  # lower bounds: 7 writers + 1, 24 tets per node
    assert g.count(capi.DISTANCE) == 5616 and g.count(capi.TET) == 4374  # SURVEY section 8 sizes


def test_bad_arguments_are_rejected():
    g = host_solver()
    g.addNodes([[0, 0, 0], [1, 0, 0]])
    with pytest.raises(capi.PiesError):
        g.add_distance([[0, 7]], 0.5)  # node id out of range
    with pytest.raises(capi.PiesError):
        g.set_schedule(5)
    L = capi.load()
    assert L.pies_count(None, 0, None) == capi.ERR_INVALID


def test_default_schedule_and_environment_override(monkeypatch):
    """pies_create starts with PIES_SCHEDULE_DEFAULT (LAYERED: the schedule bench.py's headline is measured on); PIES_SCHEDULE
    in the environment overrides it for new handles; EXACT keeps the containers in the order the host added the constraints."""
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pies_hip.h")).read()
    assert re.search(r"#define PIES_SCHEDULE_DEFAULT PIES_SCHEDULE_LAYERED", header) and capi.SCHEDULE_DEFAULT == capi.SCHEDULE_LAYERED

    def tet_order(schedule=None):
        g = capi.Solver(scenes.pbd_options(capi, 4), device=capi.DEVICE_NONE)
        scenes.build_beam(g, (6, 6, 14))
        if schedule is not None:
            g.set_schedule(schedule)
        g.finalize()
        o = g.order(capi.TET)
        g.close()
        return o
    layered, exact = tet_order(capi.SCHEDULE_LAYERED), tet_order(capi.SCHEDULE_EXACT)
    assert not np.array_equal(layered, exact)
    assert np.array_equal(tet_order(), layered)             # the default
    monkeypatch.setenv("PIES_SCHEDULE", "exact")
    assert np.array_equal(tet_order(), exact)               # the environment's choice ...
    assert np.array_equal(tet_order(capi.SCHEDULE_LAYERED), layered)  # ... does not override an explicit pies_set_schedule


def _check_tile_plan(g, pies):
    """Invariants of the PD tile plan (pd_tiles.cpp): every element pair in exactly one tile, a tile's node list is exactly the
    union of its elements' nodes (at most 128), the 8-bit local indices decode to the elements' nodes, and the per-node lists
    name every (element, corner) incidence exactly once, ascending."""
    plan = g.pd_tile_plan()
    assert plan is not None
    tets = g.ids(pies.TET).reshape(-1, 4)
    seen = np.zeros(len(tets), dtype=np.int64)
    records = 0
    for t, info in enumerate(plan["info"]):
        nn, ne = int(info & 0xffff), int(info >> 16)
        assert 1 <= ne <= 128 and 1 <= nn <= 128
        nodes, elems = plan["node"][t, :nn], plan["elem"][t, :ne]
        assert np.all(np.diff(nodes.astype(np.int64)) > 0)                      # ascending, distinct
        np.add.at(seen, elems, 1)
        loc = plan["local"][t, :ne]
        dec = np.stack([(loc >> s) & 0xff for s in (0, 8, 16, 24)], axis=1)
        assert dec.max() < nn and np.array_equal(nodes[dec], tets[elems])       # local indices name the elements' nodes
        assert set(nodes.tolist()) == set(tets[elems].reshape(-1).tolist())     # and the node list is exactly their union
        nptr = plan["nptr"][t].astype(np.int64)
        assert nptr[0] == 0 and nptr[nn] == 4 * ne and np.all(np.diff(nptr[:nn + 1]) >= 1)
        inc = plan["inc"][t, :4 * ne].astype(np.int64)
        assert sorted(inc.tolist()) == list(range(4 * ne))                      # every (element, corner) exactly once
        for k in range(nn):
            seg = inc[nptr[k]:nptr[k + 1]]
            assert np.all(np.diff(seg) > 0) and np.all(dec[seg >> 2, seg & 3] == k)
        records += nn
    assert np.all(seen == 1)
    return len(plan["info"]), records


def test_pd_tile_plan_invariants_on_a_lattice_and_on_an_unstructured_mesh(pies):
    g = pies.Solver(pies.Options(solver=pies.PD, iterations=10), device=pies.DEVICE_NONE)
    g.create_tet_box(7, 6, 23, translation=(0.0, 2.0, 0.0), w=1.0, volume=True, triangles=True)
    tiles, records = _check_tile_plan(g, pies)
    assert tiles == -(-g.count(pies.TET) // 128) and records < 5 * g.count(pies.NODES)
    mesh = scenes.delaunay_beam((6, 5, 14))
    u = pies.Solver(pies.Options(solver=pies.PD, iterations=10), device=pies.DEVICE_NONE)
    scenes.build_unstructured_pd(u, mesh)
    _check_tile_plan(u, pies)
    # a scene without strain + volume pairs keeps per-(element, node) records
    p = pies.Solver(pies.Options(solver=pies.PD, iterations=10), device=pies.DEVICE_NONE)
    p.create_tet_box(3, 3, 3, w=1.0, volume=False)
    assert p.pd_tile_plan() is None
