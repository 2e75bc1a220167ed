"""Runs INSIDE the child process of tests/test_sanitizers.py (sanitizer runtime preloaded, PIES_LIB / PIES_ORACLE_LIB pointing at
the ASan + UBSan builds).  Host logic only: scenes and plans through host-only handles (PIES_DEVICE_NONE), the oracle's ticks.
Any sanitizer report aborts the process (halt_on_error); the parent checks the exit code and the 'sanitizer child ok' line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle_api as ora  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

assert "asan" in capi.LIB_PATH and "asan" in os.environ.get("PIES_ORACLE_LIB", ""), "the sanitizer builds are not the ones loaded"


def planner_soak(n_scenes, first_seed):
    """layer_plan.cpp / schedule.cpp / wavefront.cpp: random lattices and Delaunay beams, every candidate plan, one strip and
    strips, all three schedules; the exported order must be a permutation of the container."""
    for seed in range(first_seed, first_seed + n_scenes):
        rng = np.random.default_rng(seed)
        dims = (int(rng.integers(3, 7)), int(rng.integers(3, 6)), int(rng.integers(6, 20)))
        mesh = None if rng.integers(0, 2) else scenes.delaunay_beam(dims, seed=seed, jitter=float(rng.uniform(0.1, 0.35)))
        for cand in (0, 1, 2):
            capi.set_tuning("PIES_LAYER_PLAN_FORCE", str(cand))
            strips = bool((seed + cand) % 2)
            capi.set_tuning("PIES_LAYER_ONE_STRIP_MAX", "60" if strips else "")
            capi.set_tuning("PIES_LAYER_TILE_NODES", "120" if strips else "")
            capi.set_tuning("PIES_LAYER_STRIPS_MIN_NODES", "0" if strips else "")
            for schedule in (capi.SCHEDULE_LAYERED, capi.SCHEDULE_COLOURED, capi.SCHEDULE_EXACT):
                g = capi.Solver(scenes.pbd_options(capi, 4), device=capi.DEVICE_NONE)
                if mesh is None:
                    scenes.build_beam(g, dims)
                else:
                    scenes.build_unstructured(g, mesh)
                g.add_position(np.arange(0, g.count(9), 7, dtype=np.uint32), 0.5)
                g.set_schedule(schedule)
                g.finalize()
                for t in (capi.DISTANCE, capi.TET):
                    order = g.order(t)
                    assert np.array_equal(np.sort(order), np.arange(g.count(t), dtype=np.uint32)), (seed, cand, schedule, t)
                    if schedule != capi.SCHEDULE_LAYERED:
                        g.batches(t)
                g.close()
    for name in ("PIES_LAYER_PLAN_FORCE", "PIES_LAYER_ONE_STRIP_MAX", "PIES_LAYER_TILE_NODES", "PIES_LAYER_STRIPS_MIN_NODES"):
        capi.set_tuning(name, "")


def pd_setup_soak():
    """pd_setup.cpp / pd_tiles.cpp / scene.cpp: PD scenes of every constraint kind on host-only handles, finalised (system matrix,
    SELL / windowed SELL, dictionary, tile plan)."""
    for dims, tri in (((4, 4, 9), True), ((5, 3, 14), False)):
        g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=capi.DEVICE_NONE)
        g.create_tet_box(*dims, translation=(0.0, 0.05, 0.0), w=1.0, volume=True, triangles=tri)
        g.add_position(np.arange(0, dims[0] * dims[1] * dims[2], dims[2], dtype=np.uint32), 2.0)
        g.create_shape_matching_box((12.0, 0.5, 0.0), 3, 3, 4, 2.0)
        g.create_bend_sheet(4, 5, translation=(20.0, 2.0, 0.0), scale=1.0, w=0.4)
        g.create_sheet(5, 4, translation=(30.0, 2.0, 0.0), scale=0.5, mass=2.0, w=0.3)
        region = np.eye(4, dtype=np.float32)
        region[3, :3] = (1.0, 1.0, 1.0)
        g.add_fixed_regions(region.reshape(16), 7.0)
        g.add_node_pairs(np.uint32([[0, 5], [5, 9], [2, 11]]))  # the node-pair extension container
        g.finalize()
        g.pd_tile_plan()
        g.close()
    mesh = scenes.delaunay_beam((4, 4, 10), seed=3)
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=capi.DEVICE_NONE)
    scenes.build_unstructured_pd(g, mesh)
    g.finalize()
    g.pd_tile_plan()
    g.close()


def oracle_soak():
    """The oracle's own loops: PBD with node-node collisions in both visiting rules, PD with triangle contacts and every constraint kind."""
    for rule in (0, 2):
        rng = np.random.default_rng(5)
        p = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(6), indexing="ij"), -1).reshape(-1, 3) * 0.9
        p = (p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]).astype(np.float32)
        o = ora.OracleSolver(scenes.pbd_options(ora, 4))
        o.addNodes(p)
        o.set_velocities(rng.uniform(-1, 1, p.shape).astype(np.float32))
        o.set_flag(ora.FLAG_COLLISION_RULE, rule)
        o.tick(3)
        assert np.isfinite(o.positions).all()
    o = ora.OracleSolver(scenes.pbd_options(ora, 6))
    scenes.build_beam(o, (4, 4, 8))
    o.create_bend_sheet(4, 5, translation=(20.0, 2.0, 0.0), scale=1.0, w=0.4)
    scenes.perturb(o, 2, 0.05)
    o.tick(3)
    assert np.isfinite(o.positions).all()
    o = ora.OracleSolver(ora.Options(solver=ora.PD, iterations=6))
    o.create_tet_box(4, 4, 8, translation=(0.0, 0.02, 0.0), w=1.0, volume=True, triangles=True)
    o.create_tet_box(3, 3, 3, translation=(0.6, 4.3, 2.2), w=1.0, volume=True, triangles=True)  # lands on the first
    o.create_shape_matching_box((12.0, 0.5, 0.0), 3, 3, 4, 2.0)
    o.add_node_pairs(np.uint32([[0, 5], [5, 9], [2, 140]]))
    scenes.perturb(o, 9, 0.03)
    o.set_prev_positions(o.positions)
    o.tick(4)
    assert np.isfinite(o.positions).all()


planner_soak(int(sys.argv[1]) if len(sys.argv) > 1 else 4, 1)
pd_setup_soak()
oracle_soak()
print("sanitizer child ok")
