#!/usr/bin/env python3
"""Benchmark of the Pies PBD substep on MI355X (BASELINE.json metric: substeps/sec and
constraint-projections/sec at 100k particles, HBM GB/s vs peak).

A "step" is one Solver::tick = `timeSubsteps` (1) substep of BASELINE config 2: the 20x20x250 lattice
(100 000 particles, 649 156 distance + 539 334 tet-strain constraints), PBD, 20 iterations, node-node
collisions off, synthetic perturbed rest state already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one body per GPU)

Rank 0 prints ONE JSON line.  `value` is whole-job substeps/s (N independent bodies, weak scaling; the
only collective is the timing barrier / max-reduce).  `roofline` is measured live for the dominant kernel
(k_layer: the LDS-resident sweep of schedule LAYERED) between two HIP events on the solver's stream; `cpu_baseline` is the CPU oracle
(a single-threaded restatement of the reference loop) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the CPU baseline's OpenMP threads sleep at barriers instead of spinning

import numpy as np  # noqa: E402

import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec)
# algorithmic bytes per unit (DESIGN.md "Kernels")
BYTES = {"predict": 48, "position": 44, "distance": 52, "tet": 160, "bend": 136, "floor": 20, "velocity": 40,
         "layer": 1}  # a layer launch fuses several kinds: the library tallies its units in algorithmic bytes directly
ITERATIONS = 20


def log(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    print("[bench %6.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def aggregate(elapsed_s, units, dist=None):
    """max-over-ranks time and summed units (the only communication of the whole job)."""
    if dist is None:
        return elapsed_s, units
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    u = torch.tensor([units], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def build_scene(mod, dims, seed, schedule=None, device=None):
    """The same synthetic scene through the product (mod = capi, device given) or the oracle."""
    opts = scenes.pbd_options(mod, ITERATIONS)
    s = mod.Solver(opts, device=device) if device is not None else mod.OracleSolver(opts)
    scenes.build_beam(s, dims)
    scenes.perturb(s, seed, 0.05)
    s.set_flag(1, 0)
    if schedule is not None:
        s.set_schedule(schedule)
    return s


def timed_ticks(solver, steps, warmup, barrier):
    solver.tick_async(warmup)
    solver.synchronize()
    barrier()
    t0 = time.perf_counter()
    solver.tick_async(steps)
    solver.synchronize()
    barrier()
    return time.perf_counter() - t0


def pd_bytes(solver):
    """Algorithmic bytes per unit of the PD kernels; units are constraints for the local
    steps and rows (nodes) for the rest."""
    n = solver.count(capi.NODES)
    nnz = solver.count(capi.SYSTEM_NNZ)
    paired = solver.count(capi.VOLUME) > 0 and solver.launch_counts().get("pd_local_volume", 0) == 0
    # the fused strain + volume local step leaves one record per corner of the element pair
    inc = (4 * (solver.count(capi.TET) + (0 if paired else solver.count(capi.VOLUME)) + solver.count(capi.BEND))
           + 2 * solver.count(capi.DISTANCE) + solver.count(capi.POSITION))
    # strain + volume constraints over the same elements run fused (no separate volume launches): ids 16 + Qinv 36 +
    # 2 x (min, max, w) 24 + four positions 48 + 2 x 36 projected gradients
    return {
        "pd_predict": 52, "pd_local_distance": 64, "pd_local_tet": 196 if paired else 148, "pd_local_volume": 148,
        # gather formulation: one 12-byte contribution + its 4-byte slot index per (constraint, node) incidence,
        # inertia term in, right-hand side out (the survey's scatter formulation would be 148 B per tetrahedron)
        "pd_rhs": (16.0 * inc + 32.0 * n) / n,
        "pd_spmv": (8.0 * nnz + 28.0 * n) / n,   # col + val per stored entry; rowptr, x, y per row; 3 right-hand sides fused
        "pd_cg_update": 120,                     # the PCG iteration's 10 three-component vector passes
        "pd_velocity": 60,
    }


def kernel_profile(solver, bytes_per_unit=None):
    """Per kernel class: launches per substep, average per-launch device time (us), algorithmic GB/s.
    Each class is timed in isolation by replaying a graph of only its launches (pies_profile_substep)."""
    out = {}
    lc = solver.launch_counts()
    BYTES = bytes_per_unit or globals()["BYTES"]
    for k, name in enumerate(capi.KERNEL_NAMES):
        if name not in BYTES or lc.get(name, 0) == 0:
            continue
        launches, ms, units = solver.profile_substep(k)
        if launches == 0:
            continue
        out[name] = {
            "launches_per_substep": lc[name],
            "avg_us": 1e3 * ms / launches,
            "units_per_launch": units / launches,
            "algorithmic_GBs": BYTES[name] * units / (ms * 1e-3) / 1e9 if ms > 0 else None,
        }
        out[name]["hbm_frac"] = out[name]["algorithmic_GBs"] / HBM_PEAK_GBS if ms > 0 else None
    return out


def pd_beam(dims, device):
    """BASELINE configs[2] pattern: lattice beam, Projective Dynamics, tets + volume (w = 1), 10 iterations,
    the k = 0 end cap pinned."""
    W, H, D = dims
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=device)
    g.create_tet_box(W, H, D, translation=(0.0, 2.0, 0.0), w=1.0, volume=True, triangles=True)
    g.add_position(np.array([D * (j + H * i) for i in range(W) for j in range(H)], dtype=np.uint32), 2.0)
    g.finalize()
    for _ in range(34):  # a host ticks and synchronises once per frame: the captured CG iteration budget settles to what
        g.tick_async(1)  # the solves use (two spare iterations after 8 calm frames, one after 24 more)
        g.synchronize()
    return g


def scale_profiles(device):
    """Per-kernel algorithmic bandwidth of the projection and SpMV kernels at 1M particles (100x100x100), where
    a launch is long enough for HBM rather than the kernel boundary to bound it."""
    out = {}
    for name, sched in (("pbd_1m", capi.SCHEDULE_LAYERED), ("pbd_1m_coloured", capi.SCHEDULE_COLOURED)):
        log(name)
        g = build_scene(capi, scenes.L1M, 99, schedule=sched, device=device)
        g.finalize()
        el = timed_ticks(g, 3, 1, lambda: None)
        out[name] = {"substeps_per_sec": 3 / el, "projections_per_sec": 3 / el * scenes.projections_per_substep(g, capi, ITERATIONS),
                     "launches_per_substep": sum(g.launch_counts().values()), "kernels": kernel_profile(g)}
        g.close()
    log("pd_1m")
    g = pd_beam(scenes.L1M, device)
    el = timed_ticks(g, 3, 1, lambda: None)
    out["pd_1m"] = {"substeps_per_sec": 3 / el, "pcg_stats": g.pcg_stats(), "kernels": kernel_profile(g, pd_bytes(g))}
    g.close()
    return out


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (collected
    separately, see profiles/README.md); None when no pass has been recorded for this round."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_name, {}).get("hbm_bytes_per_launch")
    except OSError:
        return None


def extra_configs(device):
    """Short secondary measurements of the other single-GPU BASELINE configs (not the headline value)."""
    out = {}
    # configs[2]: 100k beam, Projective Dynamics, tets + volume (w = 1), 10 iterations, k = 0 end cap pinned
    g = pd_beam(scenes.L100K, device)
    el = timed_ticks(g, 30, 3, lambda: None)
    res, iters, solves = g.pcg_stats()
    out["pd_config3"] = {"value": 30 / el, "unit": "substeps/s", "workload": "20x20x250 beam, PD, 539334 tet + 539334 volume constraints, "
                         "10 local/global iterations, floor contacts, Jacobi-PCG rel. tol 3e-7 (iteration budget adapts)",
                         "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters,
                         "projections_per_sec": 30 / el * 10 * (g.count(capi.TET) + g.count(capi.VOLUME) + g.count(capi.POSITION))}
    g.set_flag(capi.FLAG_TRIANGLE_COLLISIONS, 0)  # the profile pass times the tetrahedral pipeline's kernels
    g.finalize()
    out["pd_config3"]["kernels"] = kernel_profile(g, pd_bytes(g))
    g.close()
    # configs[1] on an unstructured mesh: Delaunay beam of the same size (the lattice stands in for tetgen in the headline)
    log("unstructured beam")
    mesh = scenes.delaunay_beam(scenes.L100K)
    un = {"workload": "Delaunay triangulation of a jittered 20x20x250 lattice: %d particles, %d distance + %d tet-strain constraints, "
                      "PBD, 20 iterations" % (len(mesh[0]), len(mesh[2]), len(mesh[1]))}
    for name, sched in (("layered", capi.SCHEDULE_LAYERED), ("coloured", capi.SCHEDULE_COLOURED)):
        g = capi.Solver(scenes.pbd_options(capi, ITERATIONS), device=device)
        scenes.build_unstructured(g, mesh)
        scenes.perturb(g, 1234, 0.03)
        g.set_flag(1, 0)
        g.set_schedule(sched)
        g.finalize()
        el = timed_ticks(g, 20, 2, lambda: None)
        un[name] = {"value": 20 / el, "unit": "substeps/s", "launches_per_substep": sum(g.launch_counts().values())}
        g.close()
    out["unstructured_config2"] = un
    # configs[4], one GPU's share: a 250k-particle body (25x25x400), PD, point-triangle + floor contact pipeline on
    log("config 5 share")
    g = pd_beam(scenes.L250K, device)
    el = timed_ticks(g, 20, 3, lambda: None)
    res, iters, solves = g.pcg_stats()
    out["pd_config5_per_gpu"] = {"value": 20 / el, "unit": "substeps/s", "workload": "25x25x400 beam (250000 particles), PD, tets + volume, 10 "
                                 "iterations, point-triangle CCD + floor contact pipeline on (no contact binds in this window: see pd_contacts); "
                                 "configs[4] runs one such body per GPU",
                                 "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters,
                                 "tri_contacts_last_substep": len(g.tri_collisions)}
    g.close()
    # PD with contacts that actually bind: a short beam resting on a long one that lies on the floor
    log("PD contact scene")
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=device)
    g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
    g.finalize()
    for _ in range(10):  # frame loop: the upper beam lands, the CG budget follows the contacts
        g.tick_async(1)
        g.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.tick_async(1)
        g.synchronize()
    el = time.perf_counter() - t0
    res, iters, solves = g.pcg_stats()
    out["pd_contacts"] = {"value": 10 / el, "unit": "substeps/s", "workload": "125000 particles: a 25x25x40 beam resting on a 25x25x160 beam on the "
                          "floor, PD, 10 iterations, floor + point-triangle contacts binding (w = 1e4 on the diagonal)",
                          "tri_contacts_last_substep": len(g.tri_collisions), "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters,
                          "failed": g.failed}
    g.close()
    # configs[3]: 500k loose particles, node-node collisions + floor, PBD, 4 iterations
    W, H, D = scenes.L500K
    rng = np.random.default_rng(1234)
    p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
    p = p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]
    g = capi.Solver(scenes.pbd_options(capi, 4), device=device)
    g.addNodes(p.astype(np.float32))
    g.set_velocities(np.random.default_rng(4321).uniform(-1, 1, p.shape).astype(np.float32))
    g.finalize()
    el = timed_ticks(g, 10, 2, lambda: None)
    pairs = g.collision_pairs
    out["collisions_config4"] = {"value": 10 / el, "unit": "substeps/s", "workload": "50x100x100 loose particles (r 0.5, spacing 0.9, "
                                 "jitter 0.05), PBD, 4 iterations, spatial-hash node-node collisions + floor",
                                 "resolved_pairs_per_substep": pairs / 12, "failed": g.failed}
    g.close()
    return out


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands out a
    share of a large host; OpenMP threads beyond the share only spin against each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_all_cores(dims, ticks, threads):
    """The same oracle with every host core: the containers re-ordered into the conflict-free colour classes of a
    coloured plan (built by a host-only handle: no GPU involved) and each class swept with OpenMP threads."""
    import oracle_api as ora
    plan = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_COLOURED, device=capi.DEVICE_NONE)
    plan.finalize()
    o = build_scene(ora, dims, 1234)
    for t in (capi.DISTANCE, capi.TET):
        o.permute(t, plan.order(t))
        o.set_batches(t, plan.batches(t))
    plan.close()
    o.set_threads(threads)
    t0 = time.perf_counter()
    o.tick(1)  # warm-up, and the yardstick that bounds the sample to about ten seconds
    one = time.perf_counter() - t0
    ticks = max(1, min(ticks, int(10.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    o.tick(ticks)
    dt = time.perf_counter() - t0
    return {"value": ticks / dt, "unit": "substeps/s", "cores": threads,
            "sample": "%d ticks, colour classes swept with %d OpenMP threads (bit-identical to the sequential sweep in "
                      "that order)" % (ticks, threads)}


def cpu_baseline(dims, ticks):
    import oracle_api as ora
    o = build_scene(ora, dims, 1234)
    o.tick(1)  # warm caches / page-in
    t0 = time.perf_counter()
    o.tick(ticks)
    dt = time.perf_counter() - t0
    threads = host_cores()
    return {
        "all_cores": cpu_all_cores(dims, 3 * ticks, threads),
        "value": ticks / dt, "unit": "substeps/s", "cores": 1, "kind": "port",
        "sample": "%d ticks of the same %dx%dx%d workload (20 iterations, oracle/pies_oracle.cpp, g++ -O2, 1 thread; "
                  "the reference's projection loops are single-threaded, Src/Solver.cpp:58-75)" % ((ticks,) + tuple(dims)),
        "projections_per_sec": ticks * scenes.projections_per_substep(o, ora, ITERATIONS) / dt,
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dims", type=int, nargs=3, default=list(scenes.L100K))
    ap.add_argument("--schedule", choices=["layered", "coloured", "exact"], default="layered")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra exact-order measurement")
    ap.add_argument("--cpu-ticks", type=int, default=4)
    ap.add_argument("--no-kernel-profile", action="store_true", help="skip the per-dispatch timing pass (roofline = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short PD (config 3) and collision (config 4) measurements")
    ap.add_argument("--no-scale", action="store_true", help="skip the 1M-particle per-kernel bandwidth measurements")
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("PIES_BENCH_BACKEND", "nccl")  # "gloo": test hook for boxes with fewer GPUs than ranks
        ndev = torch.cuda.device_count()
        device_index = local_rank % max(1, ndev)
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
        sync_t = torch.zeros(1, device="cuda" if backend == "nccl" else "cpu")

        def barrier():
            dist.all_reduce(sync_t)  # RCCL over xGMI: 4-byte all-reduce as the barrier
            torch.cuda.synchronize()
    else:
        device_index = local_rank

        def barrier():
            pass

    dims = tuple(args.dims)
    sched = {"layered": capi.SCHEDULE_LAYERED, "coloured": capi.SCHEDULE_COLOURED, "exact": capi.SCHEDULE_EXACT}[args.schedule]
    g = build_scene(capi, dims, 1234 + rank, schedule=sched, device=device_index)
    g.finalize()
    substeps_per_tick = g.options.timeSubsteps
    proj = scenes.projections_per_substep(g, capi, ITERATIONS)

    log("scene ready, timing %d steps" % args.steps)
    elapsed = timed_ticks(g, args.steps, args.warmup, barrier)
    elapsed, total_substeps = aggregate(elapsed, args.steps * substeps_per_tick, dist)
    assert np.isfinite(g.positions).all()

    result = None
    if rank == 0:
        value = total_substeps / elapsed
        lc = g.launch_counts()
        if args.no_kernel_profile:
            print(json.dumps({"value": value, "unit": "substeps/s", "launches_per_substep": sum(lc.values())}))
            g.close()
            return
        items = ITERATIONS * (g.count(capi.DISTANCE) + g.count(capi.TET) + g.count(capi.POSITION) + g.count(capi.BEND) + g.count(capi.NODES))
        wave_bytes = ITERATIONS * (BYTES["distance"] * g.count(capi.DISTANCE) + BYTES["tet"] * g.count(capi.TET) + BYTES["position"]
                                   * g.count(capi.POSITION) + BYTES["bend"] * g.count(capi.BEND) + BYTES["floor"] * g.count(capi.NODES)) / items
        prof = kernel_profile(g, dict(BYTES, wave=wave_bytes))  # schedule exact: a launch mixes the kinds of one dependency level
        dom = "layer" if "layer" in prof else "wave" if "wave" in prof else "tet" if "tet" in prof else max(prof, key=lambda k: prof[k]["avg_us"] * prof[k]["launches_per_substep"])
        achieved = prof[dom]["algorithmic_GBs"]
        result = {
            "metric": "substeps/sec @100k particles (PBD distance+tet-strain, 20 iterations)",
            "value": value, "unit": "substeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %dx%dx%d lattice beam, %d particles, %d distance + %d tet-strain "
                                   "constraints, PBD, %d iterations, 1 substep/tick, one body per GPU"
                                   % (dims + (g.count(capi.NODES), g.count(capi.DISTANCE), g.count(capi.TET), ITERATIONS)),
                       "schedule": args.schedule, "parallelism": "replicas x%d" % world,
                       "launches_per_substep": sum(lc.values())},
            "projections_per_sec": value * proj,
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_" + dom),
                         "avg_launch_us": prof[dom]["avg_us"],
                         "bytes_per_launch": dict(BYTES, wave=wave_bytes)[dom] * prof[dom]["units_per_launch"]},
            "kernels": prof,
        }
        # whole-substep algorithmic traffic over wall time (includes launch gaps)
        per_substep_bytes = (BYTES["predict"] + BYTES["velocity"] + ITERATIONS * BYTES["floor"]) * g.count(capi.NODES) + ITERATIONS * (
            BYTES["distance"] * g.count(capi.DISTANCE) + BYTES["tet"] * g.count(capi.TET) + BYTES["position"] * g.count(capi.POSITION)
            + BYTES["bend"] * g.count(capi.BEND))
        result["substep_algorithmic_GBs_per_gpu"] = per_substep_bytes * (value / world) / 1e9
    g.close()

    if rank == 0 and world == 1:
        if not args.no_exact and args.schedule == "layered":
            log("coloured schedule")
            c = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_COLOURED, device=device_index)
            c.finalize()
            steps = max(2, min(args.steps, 50))
            el = timed_ticks(c, steps, 2, lambda: None)
            result["coloured_schedule"] = {"value": steps * substeps_per_tick / el, "unit": "substeps/s",
                                           "launches_per_substep": sum(c.launch_counts().values()), "steps": steps,
                                           "kernels": kernel_profile(c),
                                           "note": "schedule COLOURED: one launch per colour class (24 tet + 9 distance colours per iteration)"}
            c.close()
        if not args.no_exact and args.schedule != "exact":
            log("exact schedule")
            e = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_EXACT, device=device_index)
            e.finalize()
            steps = max(2, min(args.steps, 10))
            el = timed_ticks(e, steps, 1, lambda: None)
            result["exact_order"] = {"value": steps * substeps_per_tick / el, "unit": "substeps/s",
                                     "launches_per_substep": sum(e.launch_counts().values()), "steps": steps,
                                     "note": "schedule EXACT: bit-identical to the reference's container-order sweep; one launch per level of "
                                             "the whole-substep dependency DAG"}
            e.close()
        if not args.no_extras:
            log("configs 3 and 4")
            result["other_configs"] = extra_configs(device_index)
        if not args.no_scale:
            log("1M-particle profiles")
            result["scale_1m"] = scale_profiles(device_index)
        if not args.no_cpu_baseline:
            log("CPU baseline")
            result["cpu_baseline"] = cpu_baseline(dims, args.cpu_ticks)
        log("done")
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
