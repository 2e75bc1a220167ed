#!/usr/bin/env python3
"""Benchmark of the Pies solver loop on MI355X (BASELINE.json metric: substeps/sec and constraint-projections/sec at
100k particles, HBM GB/s vs peak).

A "step" is one Solver::tick = `timeSubsteps` (1) substep of BASELINE configs[1]: the 20x20x250 lattice beam (100 000
particles, 649 156 distance + 539 334 tet-strain constraints), PBD, 20 iterations, node-node collisions off, synthetic
perturbed rest state already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one body per GPU)

Rank 0 prints ONE JSON line.  `value` is whole-job substeps/s (N independent bodies, weak scaling; the only collective is
the timing barrier / max-reduce).  How every other number is obtained:

* `roofline` (and the `roofline` of every `other_configs` entry): the dominant kernel class is timed IN SITU - whole
  substeps are launched eagerly and every launch of the class is bracketed by two HIP events on the solver's stream
  (pies_profile_in_situ), so the caches hold what the substep leaves in them; `achieved` = SURVEY 8d's algorithmic bytes of
  the launches timed / the time between their events.  `traffic` is the HBM byte count per launch of the same kernel from
  the rocprofv3 --pmc passes committed under profiles/ (`traffic_source` names the file; collected by
  tools/profile_round.sh, never inside this run).
* `cpu_baseline`: the CPU oracle (oracle/pies_oracle.cpp, a single-threaded restatement of the reference loop with the
  reference's own threading where it has any), built -O3 -march=native ON this host, timed on a bounded sample.
* `tick_inclusive`: the same workload through pies_tick (one pinned D2H copy of the positions per tick + the host-side
  unpack a C++ host pays) and through the double-buffered asynchronous export.
* `exact_order` / `coloured_schedule` / `order_deviation`: the other schedules' throughput and how far their results are
  from the reference order (EXACT).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "benchlib")):
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
K = {name: i for i, name in enumerate(capi.KERNEL_NAMES)}
DEVICE_KERNEL = {"layer": "k_layer", "tet": "k_tet", "wave": "k_wave", "pd_local_tet": "k_pd_local_tet_pair", "pd_spmv": "k_cg_ap",
                 "pd_rhs": "k_pd_rhs", "pd_cg_update": "k_cg_update", "collide": "k_pair_round4", "hash": "k_radix_scatter"}


def device_kernel(solver, cls):
    """Name of the kernel that runs class `cls` in the graph variant `solver` captured."""
    if cls == "pd_local_tet" and solver.count(capi.PD_TILES):
        return "k_pd_local_tiles"
    if solver.count(capi.PD_CG_SINGLE):
        if cls == "pd_spmv":
            return "k_cg1_iter"
        if cls == "pd_rhs" and solver.count(capi.PD_TILES):
            return "k_cg1_init"
    return DEVICE_KERNEL.get(cls, "k_" + cls)


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


# ---- rooflines ---------------------------------------------------------------------------------------------------------
def pmc_traffic(kernel_name, workload):
    """HBM bytes per launch of `kernel_name` in `workload` from the committed rocprofv3 --pmc passes (separate FETCH_SIZE /
    WRITE_SIZE passes over tools/profile_target.py <workload>, gfx950 correction (2 FETCH + WRITE) * 1024, see
    profiles/README.md); (None, None) when no pass was recorded."""
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(workload, {}).get(kernel_name, {}).get("hbm_bytes_per_launch")
            if v is not None:
                return v, "profiles/%s [%s][%s]" % (name, workload, kernel_name)
        except (OSError, ValueError, AttributeError):
            pass
    return None, None


def pmc_traffic_of_pass(prefix, workload, anchor):
    """HBM bytes per PASS of a multi-kernel stage (the pair-ordered resolve: k_pair_*; the grid rebuild: k_grid_*, k_scan_*,
    k_radix_* - `prefix` may be a tuple): the sum over its kernels of (bytes per launch x launches) in the committed --pmc
    passes, per launch of `anchor` (a kernel that runs once per pass)."""
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                w = json.load(f).get(workload, {})
            passes = w.get(anchor, {}).get("dispatches")
            if passes:
                total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for k, v in w.items() if k.startswith(prefix))
                label = prefix if isinstance(prefix, str) else " + ".join(p + "*" for p in prefix)
                return total / passes, "profiles/%s [%s][%s] per %s" % (name, workload, label, anchor)
        except (OSError, ValueError, AttributeError, KeyError, TypeError):
            pass
    return None, None


def pmc_valu(kernel_name, workload):
    """SQ_INSTS_VALU (wave-instructions) per launch of `kernel_name` from the committed rocprofv3 --pmc pass of the workload"""
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_valu_pd.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(workload, {}).get(kernel_name, {}).get("SQ_INSTS_VALU")
            if v:
                return v, "profiles/%s [%s][%s]" % (name, workload, kernel_name)
        except (OSError, ValueError, AttributeError):
            pass
    return None, None


def rocprof_average(kernel_name, workload):
    """Average duration (us) of `kernel_name` in the committed rocprofv3 --kernel-trace --stats summary of the workload."""
    import csv
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_%s_kernel_stats.csv" % workload)), reverse=True):
        tot, calls = 0.0, 0
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
            short = r["Name"].split("(")[0].replace("void ", "").split("::")[-1].split("<")[0]
            if short == kernel_name:
                tot += float(r["TotalDurationNs"])
                calls += int(r["Calls"])
        if calls:
            return tot / calls / 1e3, "profiles/" + name
    return None, None


def roofline(solver, cls, bytes_per_unit, substeps=3, note=None, workload="config2", whole_graph=False):
    """Roofline block of kernel class `cls` (a name of capi.KERNEL_NAMES), timed live with HIP events on the solver's stream.
    whole_graph: the captured substep consists of launches of this class only (config 2 under schedule LAYERED): the class is
    timed as the replay of that very graph between two events (duration / launches).  Otherwise in situ: every launch of the
    class inside eagerly launched whole substeps is bracketed by two events, and what a bracket costs around nothing (measured
    in the same pass) is taken off."""
    kname = device_kernel(solver, cls)
    if whole_graph:
        launches, ms, units = solver.profile_substep(K[cls])
        if launches == 0 or ms <= 0:
            return None
        nbytes = bytes_per_unit * units
        achieved = nbytes / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(kname, workload)
        ref, refsrc = rocprof_average(kname, workload)
        out = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": traffic, "traffic_source": src, "avg_launch_us": 1e3 * ms / launches, "launches_timed": launches,
               "bytes_per_launch": nbytes / launches, "rocprofv3_avg_us": ref, "rocprofv3_source": refsrc,
               "method": "two HIP events on the solver's stream around 5 replays of the captured substep graph, which holds launches of "
                         "this kernel only: duration / launches (kernel boundaries included)"}
        if note:
            out["note"] = note
        if ref:
            out["frac_rocprofv3"] = nbytes / launches / (ref * 1e-6) / 1e9 / HBM_PEAK_GBS
        return primary_from_profile(out)
    launches, ms, units, overhead_ms = solver.profile_in_situ(K[cls], substeps)
    if launches == 0 or ms <= 0:
        return None
    nbytes = bytes_per_unit * units
    net_ms = ms - launches * overhead_ms  # the brackets' own cost, calibrated in the same pass, taken off
    clamped = net_ms < 0.25 * ms  # the bracket is mostly overhead: no bandwidth claim from such a difference
    if clamped:
        net_ms = ms
    achieved = nbytes / (net_ms * 1e-3) / 1e9
    stage = {"collide": (("k_pair_",), "k_pair_save", "k_pair_* (one pass)", "k_pair_round"),
             "hash": (("k_grid_", "k_scan_", "k_radix_"), "k_grid_range", "k_grid_* + k_scan_* + k_radix_* (one rebuild)", "k_radix_scatter")}.get(cls)
    traffic, src = pmc_traffic_of_pass(stage[0], workload, stage[1]) if stage else pmc_traffic(kname, workload)
    if traffic is not None and traffic < 0.01 * nbytes / launches:
        # the committed --pmc pass averaged launches most of which returned at once (CG kernels of a solve that had converged)
        traffic, src = None, "%s: average over launches that mostly exit at once - not a figure for the working launch" % src
    out = {"bound": "hbm", "kernel": stage[2] if stage else kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": src, "avg_launch_us": 1e3 * net_ms / launches, "launches_timed": launches,
           "bytes_per_launch": nbytes / launches, "avg_bracket_us": 1e3 * ms / launches, "bracket_overhead_us": 1e3 * overhead_ms,
           "overhead_clamped": clamped,
           "rocprofv3_avg_us": rocprof_average(kname, workload)[0], "rocprofv3_source": rocprof_average(kname, workload)[1],
           "method": "HIP events on the solver's stream around every launch of the class inside %d eagerly launched whole substeps "
                     "(pies_profile_in_situ); avg_launch_us = the bracket minus what a bracket costs around nothing (two event packets "
                     "and the end-of-kernel write-back wait, measured in the same pass around an empty kernel)" % substeps}
    if note:
        out["note"] = note
    if out.get("rocprofv3_avg_us") and out["rocprofv3_avg_us"] >= 0.5 * out["avg_launch_us"]:
        # the same bytes over the committed rocprofv3 trace's average duration of the kernel (the pessimistic clock: the profiler's own
        # overhead is in it, the bracket correction of the in-situ figure is not).  Not from a trace whose launches are mostly early
        # exits (an average far below the working launch's time).
        out["frac_rocprofv3"] = nbytes / launches / (out["rocprofv3_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    return primary_from_profile(out)


def valu_issue_of_layer(duration_us):
    """The OTHER roofline of k_layer at 100k particles: a launch with fewer tiles than compute units lasts as long
    as its busiest wavefront's instruction stream, and a wavefront alone on its SIMD issues one VALU instruction per 4 cycles
    (MI355X_MICROARCH.md).  Fraction = that wavefront's VALU instructions (counted in the ISA: profiles/*_layer_critical_path.json)
    x 4 cycles / 2.4 GHz / the launch's duration: the share of the launch in which the critical wavefront is issuing arithmetic."""
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_layer_critical_path.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                c = json.load(f)
            insts = c["tet_colour_valu"] * c["tet_colours"] + c["distance_colour_valu"] * c["distance_colours"] + c["load_store_valu"]
            us = insts * c["cycles_per_valu_one_wavefront_per_simd"] / 2.4e3
            return {"critical_wavefront_valu": insts, "critical_wavefront_issue_us": us, "frac": us / duration_us, "source": "profiles/" + name}
        except (OSError, ValueError, KeyError):
            pass
    return None


def primary_from_profile(out):
    """`frac` / `achieved` are the figures anyone can recompute from the committed profile: bytes per launch / the kernel's AverageNs
    in profiles/*_kernel_stats.csv / 8 TB/s.  The live measurement of this run moves to frac_in_situ / achieved_in_situ (it is 10-15 %
    higher where both exist: its event brackets are corrected for their own cost, rocprofv3's durations are not).  Without a
    committed profile of the kernel the live figure stays the primary one and `frac_source` says so."""
    out["frac_in_situ"], out["achieved_in_situ"] = out["frac"], out["achieved"]
    if out.get("frac_rocprofv3"):
        out["frac"] = out["frac_rocprofv3"]
        out["achieved"] = out["frac"] * HBM_PEAK_GBS
        out["frac_source"] = "bytes_per_launch / rocprofv3_avg_us (%s) / peak" % out.get("rocprofv3_source")
    else:
        out["frac_source"] = "live HIP events of this run (no committed rocprofv3 trace of this kernel in this workload)"
    return out


def pd_bytes(solver):
    """Algorithmic bytes per unit of the PD kernels; units are constraints for the local steps and rows
    (nodes) for the rest."""
    n = solver.count(capi.NODES)
    nnz = solver.count(capi.SYSTEM_NNZ)
    paired = solver.count(capi.VOLUME) > 0 and solver.launch_counts().get("pd_local_volume", 0) == 0
    tiles, records = solver.count(capi.PD_TILES), solver.count(capi.PD_TILE_RECORDS)
    ntet = solver.count(capi.TET)
    inc = (4 * (ntet + (0 if paired else solver.count(capi.VOLUME)) + solver.count(capi.BEND))
           + 2 * solver.count(capi.DISTANCE) + solver.count(capi.POSITION))
    # 148 B per tet / volume projection.  A fused strain + volume launch does two projections per element from
    # ONE gather and adds the two contributions into one 12-byte record per corner: ids 16 + Qinv 36 + 2 x (min, max, w) 24
    # + four positions 48 + four records 48 = 172 B; with the rest dictionary the 60 bytes of constants are a 2-byte index
    # into a cache-resident table: 114 B.
    local = (114 if solver.count(capi.REST_SETS) else 172) if paired else 148
    if tiles and paired:
        # tile-resident (round 4): per element pair four 8-bit node indices 4 + its four list entries 8 + constants (2 with the
        # dictionary, 64 without); per (tile, node) record: node index 4 + position 16 + list offset 2 + the sum written 12
        local = 12 + (2 if solver.count(capi.REST_SETS) else 64) + 34.0 * records / max(1, ntet)
        inc = inc - 4 * ntet + records  # what the right-hand side gathers: one sum per (tile, node)
    rhs = (16.0 * inc + 32.0 * n) / n
    single = bool(solver.count(capi.PD_CG_SINGLE))
    return {
        "pd_local_tet": local, "pd_local_volume": 148, "pd_local_distance": 64, "pd_predict": 52,
        # gather formulation: one 12-byte contribution + its 4-byte slot index per incidence, inertia term in, right-hand side
        # out.  When the residual kernel of the one-launch CG evaluates the right-hand side itself (tiles), the launch also does
        # the residual's SpMV: + 8 nnz + 28 N (and the right-hand side is never written)
        "pd_rhs": rhs + ((8.0 * nnz + 28.0 * n) / n - 16.0 if single and tiles else 0.0),
        # SpMV 8 nnz + 28 N; a whole PCG iteration (SpMV + its ten three-component vector passes) 8 nnz + 148 N.
        # k_cg1_iter IS a whole iteration in one launch; k_cg_ap is the SpMV (+ the direction update) of the two-launch form
        "pd_spmv": (8.0 * nnz + (148.0 if single else 28.0) * n) / n,
        "pd_spmv_only": (8.0 * nnz + 28.0 * n) / n,
        # what the CG-iteration launch has to move in the form it runs in: the ten vector passes (148 N) + the matrix - one word per
        # row with the row dictionary (the stencils stay in the vector cache), 6 B per stored entry (value + 16-bit window slot, SELL
        # padding included) + 2 B per halo column with the windowed matrix, 8 B per entry with the SELL arrays
        "pd_spmv_required": (148.0 * n + (4.0 * n if solver.count(capi.ROW_STENCILS) and not solver.count(capi.PD_WINDOW_ENTRIES) else
                                          6.0 * solver.count(capi.PD_WINDOW_ENTRIES) + 2.0 * solver.count(capi.PD_WINDOW_HALO)
                                          if solver.count(capi.PD_WINDOW_ENTRIES) else 8.0 * nnz)) / n,
        "pd_cg_update": 120,                     # the PCG iteration's 10 three-component vector passes
        "pd_velocity": 60,
    }


def pd_rooflines(g, workload, substeps):
    """The three PD rooflines of a solver in its captured variant: local step, the CG iteration (by SURVEY 8d's bytes of what the
    launch does AND by the SpMV's bytes alone), right-hand side."""
    B = pd_bytes(g)
    single, tiles = bool(g.count(capi.PD_CG_SINGLE)), bool(g.count(capi.PD_TILES))
    out = {"roofline": roofline(g, "pd_local_tet", B["pd_local_tet"], substeps=substeps, workload=workload, note=(
        "tile-resident strain + volume local step (one wavefront per tile of 128 element pairs, two per lane in packed fp32): the "
        "bytes are what the launch has to move - 12-14 B per element pair + 34 B per (tile, node) sum; SURVEY 8d's count for the "
        "same work would be two 148-B projections + the 148-B scatter into the right-hand side per element.  The launch is bound "
        "by VALU issue (about 2 000 instructions per lane), not by HBM" if tiles else
        "fused strain + volume local step: two projections per element from one gather and one SVD, one 12-byte record per "
        "corner; bound by VALU issue at 100k particles"))}
    rl = out["roofline"]
    if rl:
        # the yardstick of a launch bound by VALU issue: wave-instructions x 2 cycles (a SIMD issues a wave64 VALU instruction every 2
        # cycles when it has two wavefronts to take them from, MI355X_MICROARCH.md "Per-instruction cycle constants") over what the
        # chip's 1 024 SIMDs offer in the launch's duration at 2.4 GHz
        insts, src = pmc_valu(rl["kernel"], workload)
        if insts:
            rl["valu_wave_instructions_per_launch"] = insts
            rl["valu_source"] = src
            rl["valu_frac"] = insts * 2.0 / (1024 * 2.4e9 * rl["avg_launch_us"] * 1e-6)
    # (the committed trace of the CG iteration's WORKING launches, where there is one: tools/profile_target.py <workload>_work runs the
    # same scene with the converged exit off - in the plain trace of a body at rest the kernel's launches are early exits)
    work = workload + "_work" if rocprof_average(device_kernel(g, "pd_spmv"), workload + "_work")[0] else workload
    sp = roofline(g, "pd_spmv", B["pd_spmv"], substeps=substeps, workload=work, note=(
        "ONE launch = one whole PCG iteration (Chronopoulos-Gear form: scalars from the previous launch's partial sums, the "
        "neighbours' new preconditioned residual recomputed in the gather, x / r / p / s of the own rows, the next dot products): "
        "SURVEY 8d's 8 nnz + 148 N bytes per launch; frac_spmv_bytes_only prices the same launch by the SpMV's 8 nnz + 28 N alone.  "
        "The solves of the timed pass do not take the converged early exit.  On a lattice the launch does not stream the matrix: "
        "rows with the same stencil share one copy of it (row dictionary)" if single else
        "SELL-64 SpMV over 3 right-hand sides + fused direction update; 8 nnz + 28 N bytes per launch (SURVEY 8d)"))
    if sp:
        sp["frac_spmv_bytes_only"] = sp["frac"] * B["pd_spmv_only"] / B["pd_spmv"]
        # the same launch priced by the bytes its form of the matrix really has to move (the row dictionary streams no matrix)
        sp["frac_required_bytes"] = sp["frac"] * B["pd_spmv_required"] / B["pd_spmv"]
        sp["required_bytes_per_row"] = B["pd_spmv_required"]
        if g.count(capi.ROW_STENCILS) and not g.count(capi.PD_WINDOW_ENTRIES):
            # the row dictionary streams no matrix: SURVEY 8d's 8 nnz + 148 N would price bytes the launch never moves.  `frac` is
            # the fraction by the bytes this form has to move; the survey's pricing stays as frac_survey_bytes
            for k in ("frac", "frac_in_situ", "frac_rocprofv3", "achieved", "achieved_in_situ"):
                if sp.get(k) is not None:
                    sp[k + "_survey_bytes"] = sp[k]
                    sp[k] = sp[k] * B["pd_spmv_required"] / B["pd_spmv"]
            sp["frac_required_bytes"] = sp["frac"]
            sp["bytes_per_launch_survey"] = sp["bytes_per_launch"]
            sp["bytes_per_launch"] = sp["bytes_per_launch"] * B["pd_spmv_required"] / B["pd_spmv"]
        sp["matrix_form"] = ("row dictionary (%d stencils)" % g.count(capi.ROW_STENCILS) if g.count(capi.ROW_STENCILS) and not g.count(capi.PD_WINDOW_ENTRIES)
                             else "windowed SELL (%.3f stored entries per matrix entry, %.2f halo columns per row)" % (
                                 g.count(capi.PD_WINDOW_ENTRIES) / max(1, g.count(capi.SYSTEM_NNZ)), g.count(capi.PD_WINDOW_HALO) / max(1, g.count(capi.NODES)))
                             if g.count(capi.PD_WINDOW_ENTRIES) else "SELL-64")
    out["roofline_spmv"] = sp
    out["roofline_rhs"] = roofline(g, "pd_rhs", B["pd_rhs"], substeps=substeps, workload=workload, note=(
        "the residual kernel of the one-launch CG, which also evaluates the right-hand side (inertia term + the node's 3-4 tile sums "
        "+ contact / goal / floor terms): there is no k_pd_rhs launch and the right-hand side never travels through HBM" if single and tiles
        else None))
    return out


def replay_latencies(solver, bytes_per_unit=None):
    """Per kernel class: launches per substep and the average per-launch time of an ISOLATED replay (a graph holding only
    that class's launches, pies_profile_substep).  One class's working set is usually cache resident in such a replay, so
    these are launch-latency figures (kernel boundary + dependent memory round trips), not bandwidth figures."""
    out = {}
    lc = solver.launch_counts()
    for k, name in enumerate(capi.KERNEL_NAMES):
        if lc.get(name, 0) == 0 or name in ("hash", "collide"):
            continue
        launches, ms, units = solver.profile_substep(k)
        if launches == 0:
            continue
        out[name] = {"launches_per_substep": lc[name], "isolated_replay_avg_us": 1e3 * ms / launches, "units_per_launch": units / launches}
    return out


# ---- scenes of the other configs ----------------------------------------------------------------------------------------
def pd_beam(dims, device, settle=34, mod=capi, pcg=None):
    """BASELINE configs[2] pattern: lattice beam, Projective Dynamics, tets + volume (w = 1), 10 iterations, the k = 0 end
    cap pinned."""
    W, H, D = dims
    opts = mod.Options(solver=mod.PD, iterations=10)
    g = mod.Solver(opts, device=device) if mod is capi else mod.OracleSolver(opts)
    g.create_tet_box(W, H, D, translation=(0.0, 2.0, 0.0), w=1.0, volume=True, triangles=True)
    g.add_position(np.array([D * (j + H * i) for i in range(W) for j in range(H)], dtype=np.uint32), 2.0)
    if mod is capi:
        if pcg:
            g.set_pcg(*pcg)
        g.finalize()
        for _ in range(settle):  # a host ticks and synchronises once per frame: the captured CG iteration budget settles to
            g.tick_async(1)      # what the solves use (two spare iterations after 8 calm frames, one after 24 more)
            g.synchronize()
    return g


def config4_particles():
    W, H, D = scenes.L500K
    rng = np.random.default_rng(1234)
    p = np.stack(np.meshgrid(np.arange(W), np.arange(H), np.arange(D), indexing="ij"), -1).reshape(-1, 3) * 0.9
    p = p + rng.uniform(-0.05, 0.05, p.shape) + [0, 0.5, 0]
    v = np.random.default_rng(4321).uniform(-1, 1, p.shape)
    return p.astype(np.float32), v.astype(np.float32)


def contact_scene(mod, device=None, dims=None):
    """One GPU's share of BASELINE configs[4] with contacts that bind: the 25x25x400 body (250 000 particles) lying on the
    floor and a second, small body landing on it (the scene of tests/test_tri_collisions_gpu.py's config-5 parity test)."""
    opts = mod.Options(solver=mod.PD, iterations=10)
    g = mod.Solver(opts, device=device) if mod is capi else mod.OracleSolver(opts)
    W, H, D = dims or scenes.L250K
    g.create_tet_box(W, H, D, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(8, 6, 30, translation=(3.3, 0.04 + (H - 1) + 0.04, 40.4), w=1.0, volume=True, triangles=True)
    v = g.velocities
    v[W * H * D:, 1] = -2.0
    g.set_velocities(v)
    g.set_prev_positions(g.positions)
    return g


def frame_spread(frames, contacts):
    """Largest frame over the median of the frames of ITS OWN regime (point-triangle contacts binding in the frame or not): the
    measure of stalls - a graph instantiated in a frame, a solve run again - in a scene whose two regimes differ by the work they
    do (a frame with thousands of binding contacts runs six times the CG iterations of a contact-free one and the sequential
    contact passes).  Also the ratio of the two regimes' medians.  Regimes with fewer than three frames are left out."""
    worst, med = 0.0, {}
    for regime in (True, False):
        fs = sorted(f for f, c in zip(frames, contacts) if (c > 0) == regime)
        if len(fs) >= 3:
            med[regime] = fs[len(fs) // 2]
            worst = max(worst, fs[-1] / med[regime])
    return (worst or None), (med[True] / med[False] if len(med) == 2 else None)


def frame_loop(g, frames):
    """substeps/s of a host that synchronises once per frame (the CG budget follows the contacts)"""
    t0 = time.perf_counter()
    for _ in range(frames):
        g.tick_async(1)
        g.synchronize()
    return frames * g.options.timeSubsteps / (time.perf_counter() - t0)


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands out a share of
    a large host; OpenMP threads beyond the share only spin against each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def oracle_module():
    """The CPU oracle built -O3 -march=native on THIS host (falls back to the portable build if that compile fails)."""
    os.environ["PIES_ORACLE_NATIVE"] = "1"
    import oracle_api as ora
    try:
        ora.lib()
        flags = "g++ -O3 -march=native -ffp-contract=off"
    except Exception as e:  # noqa: BLE001
        log("native oracle build failed (%s): portable build" % e)
        os.environ["PIES_ORACLE_NATIVE"] = "0"
        ora.lib()
        flags = "g++ -O3 -march=x86-64-v3 -ffp-contract=off"
    return ora, flags


def timed_oracle_ticks(o, budget_s, max_ticks):
    """Bounded sample: one warm tick (page-in; also the yardstick), then as many ticks as fit the budget."""
    t0 = time.perf_counter()
    o.tick(1)
    one = time.perf_counter() - t0
    ticks = max(1, min(max_ticks, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    o.tick(ticks)
    return ticks, time.perf_counter() - t0


def cpu_all_cores(ora, dims, threads, budget_s):
    """The same oracle with every host core: the containers re-ordered into the conflict-free colour classes of a coloured
    plan (built by a host-only handle: no GPU involved) and each class swept with OpenMP threads."""
    plan = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_COLOURED, device=capi.DEVICE_NONE)
    plan.finalize()
    o = build_scene(ora, dims, 1234)
    for t in (capi.DISTANCE, capi.TET):
        o.permute(t, plan.order(t))
        o.set_batches(t, plan.batches(t))
    plan.close()
    o.set_threads(threads)
    ticks, dt = timed_oracle_ticks(o, budget_s, 12)
    return {"value": ticks / dt, "unit": "substeps/s", "cores": threads,
            "sample": "%d ticks, colour classes swept with %d OpenMP threads (bit-identical to the sequential sweep in that order)" % (ticks, threads)}


def cpu_baseline(dims, budget_s, with_all_cores=True):
    ora, flags = oracle_module()
    o = build_scene(ora, dims, 1234)
    ticks, dt = timed_oracle_ticks(o, budget_s, 8)
    out = {
        "value": ticks / dt, "unit": "substeps/s", "cores": 1, "kind": "port",
        "sample": "%d ticks of the same %dx%dx%d workload (20 iterations; oracle/pies_oracle.cpp, %s, built on this host; 1 thread: "
                  "the reference's projection loops are single-threaded, Src/Solver.cpp:58-75)" % ((ticks,) + tuple(dims) + (flags,)),
        "projections_per_sec": ticks * scenes.projections_per_substep(o, ora, ITERATIONS) / dt,
        "host_cpus": os.cpu_count(),
    }
    if with_all_cores:
        out["all_cores"] = cpu_all_cores(ora, dims, host_cores(), budget_s)
    return out


def cpu_baseline_pd(dims, budget_s):
    ora, flags = oracle_module()
    o = pd_beam(dims, None, mod=ora)
    o.set_reference_threads(True)
    ticks, dt = timed_oracle_ticks(o, budget_s, 4)
    return {"value": ticks / dt, "unit": "substeps/s", "cores": 8, "kind": "port",
            "sample": "%d ticks of the same PD workload (%s); threading as the reference: threadCount = 8 threads for collision detection "
                      "(Solver.h:36), everything else 1 thread; the global step is a banded fp32 Cholesky re-factored every substep "
                      "(bandwidth 421; Eigen's SimplicialLLT of the reference is a sparse factorisation with fewer operations: this "
                      "figure is a lower bound of the reference's speed)" % (ticks, flags)}


def cpu_baseline_collisions(budget_s):
    ora, flags = oracle_module()
    p, v = config4_particles()
    o = ora.OracleSolver(scenes.pbd_options(ora, 4))
    o.addNodes(p)
    o.set_velocities(v)
    o.set_reference_threads(True)  # 16 insert threads, each scanning every node (SpatialHash.h:134-176)
    ticks, dt = timed_oracle_ticks(o, budget_s, 4)
    return {"value": ticks / dt, "unit": "substeps/s", "cores": min(16, host_cores()), "kind": "port",
            "sample": "%d ticks of the same 500k-particle workload (%s); threading as the reference: 16 threads for the hash insert "
                      "(SpatialHash.h:134), the resolve loop 1 thread (Solver.cpp:85-130); reference collision order" % (ticks, flags)}


def cpu_baseline_default_tick(dims, budget_s):
    """the distance-only 100k lattice as Solver::tickPBD runs it by default: node-node pass ON, reference order, the reference's 16 insert threads."""
    ora, flags = oracle_module()
    o = ora.OracleSolver(scenes.pbd_options(ora, ITERATIONS))
    scenes.build_beam(o, dims, tets=False)
    scenes.perturb(o, 1234, 0.05)
    o.set_flag(1, 1)
    o.set_reference_threads(True)
    ticks, dt = timed_oracle_ticks(o, budget_s, 3)
    return {"value": ticks / dt, "unit": "substeps/s", "cores": min(16, host_cores()), "kind": "port",
            "sample": "%d ticks, node-node pass on, reference order; 16 insert threads (SpatialHash.h:134), the rest 1 thread (%s)" % (ticks, flags)}


# ---- secondary measurements --------------------------------------------------------------------------------------------
def tick_inclusive(dims, device, sched, steps):
    """What a host pays per tick beyond the resident figure: pies_tick (pinned D2H of the positions + unpack into the
    host mirror + the strided write into a Vertex array, i.e. Pies::Solver::tick) and the asynchronous export pipeline
    (pies_tick_begin / pies_export_acquire: frame k travels while frame k+1 computes)."""
    g = build_scene(capi, dims, 1234, schedule=sched, device=device)
    g.finalize()
    g.tick(3)
    t0 = time.perf_counter()
    for _ in range(steps):
        g.tick()
        g.read_positions_strided(9)
    sync = steps / (time.perf_counter() - t0)
    frames = [g.tick_begin()]
    t0 = time.perf_counter()
    for _ in range(steps):
        frames.append(g.tick_begin())
        f = frames.pop(0)
        g.export_acquire(f)
        g.export_release(f)
    asyn = steps / (time.perf_counter() - t0)
    g.export_acquire(frames[0])
    g.export_release(frames[0])
    g.close()
    return {"pies_tick_substeps_per_sec": sync, "async_export_substeps_per_sec": asyn, "steps": steps,
            "note": "pies_tick: one 1.6 MB pinned D2H copy + host unpack + strided write into a 36-byte Vertex stream per tick "
                    "(what Pies::Solver::tick does); async export: the copy of frame k overlaps the kernels of frame k+1"}


def order_deviation(device):
    import deviation
    out = {}
    names = {capi.SCHEDULE_COLOURED: "coloured", capi.SCHEDULE_LAYERED: "layered"}

    def beam(dims, iters, tets=True):
        def make(schedule):
            g = capi.Solver(scenes.pbd_options(capi, iters), device=device)
            scenes.build_beam(g, dims, tets=tets)
            scenes.perturb(g, 1234, 0.05)
            g.set_flag(capi.FLAG_NODE_COLLISIONS, 0)
            g.set_schedule(schedule)
            return g
        d = deviation.compare(make, capi, [capi.SCHEDULE_EXACT, capi.SCHEDULE_COLOURED, capi.SCHEDULE_LAYERED])
        return {names[k]: v for k, v in d.items()}
    out["config1_l1k_10_iterations"] = beam(scenes.L1K, 10)
    # the same lattice with the distance constraints alone (createBox): a relaxation that converges, for scale
    out["l1k_distance_constraints_only_10_iterations"] = beam(scenes.L1K, 10, tets=False)
    log("order deviation: config 2")
    out["config2_l100k_20_iterations"] = beam(scenes.L100K, ITERATIONS)
    log("order deviation: collisions")
    particles = scenes.loose_particles
    p, v = particles((25, 50, 50))  # config 4 at 1/8 (the reference-order pass is one sequential chain, ~20 us per node)

    def make(rule):
        g = capi.Solver(scenes.pbd_options(capi, 4), device=device)
        g.addNodes(p)
        g.set_velocities(v)
        g.set_flag(capi.FLAG_REFERENCE_COLLISION_ORDER, rule == 0)
        return g
    out["config4_at_62500_particles_parallel_vs_reference_collision_order"] = deviation.compare(make, capi, [0, 1], ticks=(1, 5))[1]
    out["note"] = ("every schedule sweeps the same constraints with the same arithmetic; EXACT is the reference's order (containers in "
                   "insertion order, colliding nodes in ascending index). max_abs_dpos / centre_of_mass_delta are against EXACT on the "
                   "same inputs; residuals = RMS constraint violation of each result (distance: |len - rest|; tet: singular values of F "
                   "outside [0.8, 1]). Report, not a gate (SURVEY 8c).")
    return out


def run_config3(device, full):
    """configs[2]: 100k beam, Projective Dynamics, tets + volume (w = 1), 10 iterations, k = 0 end cap pinned"""
    g = pd_beam(scenes.L100K, device)
    el = timed_ticks(g, 30, 3, lambda: None)
    res, iters, solves = g.pcg_stats()
    out = {"value": 30 / el, "unit": "substeps/s", "workload": "BASELINE configs[2]: 20x20x250 beam, PD, 539334 tet + 539334 volume "
           "constraints, 10 local/global iterations, floor + point-triangle pipeline on, Jacobi-PCG rel. tol 3e-7 (captured "
           "iteration budget adapts)", "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters, "pcg_health": g.pcg_health(),
           "launches_per_substep": sum(g.launch_counts().values()), "tiles": g.count(capi.PD_TILES),
           "tile_records_per_node": g.count(capi.PD_TILE_RECORDS) / max(1, g.count(capi.NODES)), "cg_one_launch_per_iteration": bool(g.count(capi.PD_CG_SINGLE)),
           "projections_per_sec": 30 / el * 10 * (g.count(capi.TET) + g.count(capi.VOLUME) + g.count(capi.POSITION))}
    out.update(pd_rooflines(g, "config3", 3))
    if full:
        out["isolated_replay_latencies"] = replay_latencies(g)
    g.close()
    return out


def run_unstructured(device, with_coloured=True):
    """configs[1] and configs[2] on an unstructured mesh: Delaunay beam of the same size (the lattice stands in for tetgen in the
    headline): PBD under LAYERED (and COLOURED), and PD with its local-step / CG-iteration / right-hand-side rooflines on the
    paths a lattice does not take (SELL matrix instead of the row dictionary, per-element rest constants)."""
    mesh = scenes.delaunay_beam(scenes.L100K)
    un = {"workload": "Delaunay triangulation of a jittered 20x20x250 lattice: %d particles, %d distance + %d tet-strain constraints, "
                      "PBD, 20 iterations" % (len(mesh[0]), len(mesh[2]), len(mesh[1]))}
    for name, sched in (("layered", capi.SCHEDULE_LAYERED), ("coloured", capi.SCHEDULE_COLOURED)):
        if name == "coloured" and not with_coloured:
            continue
        g = capi.Solver(scenes.pbd_options(capi, ITERATIONS), device=device)
        scenes.build_unstructured(g, mesh)
        scenes.perturb(g, 1234, 0.03)
        g.set_flag(1, 0)
        g.set_schedule(sched)
        g.finalize()
        el = timed_ticks(g, 20, 2, lambda: None)
        un[name] = {"value": 20 / el, "unit": "substeps/s", "launches_per_substep": sum(g.launch_counts().values())}
        g.close()
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=device)
    scenes.build_unstructured_pd(g, mesh)
    g.finalize()
    for _ in range(20):
        g.tick_async(1)
        g.synchronize()
    el = timed_ticks(g, 20, 2, lambda: None)
    res, iters, solves = g.pcg_stats()
    pd = un["pd"] = {"value": 20 / el, "unit": "substeps/s", "workload": "the same mesh, PD: a strain + a volume constraint per tetrahedron (w = 1), "
                     "surface triangles, end cap pinned, 10 local/global iterations", "pcg_max_rel_residual": res,
                     "pcg_max_iterations_used": iters, "pcg_health": g.pcg_health(), "failed": g.failed,
                     "rest_sets": g.count(capi.REST_SETS), "row_stencils": g.count(capi.ROW_STENCILS), "tiles": g.count(capi.PD_TILES),
                     "tile_records_per_node": g.count(capi.PD_TILE_RECORDS) / max(1, g.count(capi.NODES)),
                     "launches_per_substep": sum(g.launch_counts().values())}
    pd.update(pd_rooflines(g, "none", 2))
    g.close()
    return un


def run_config5_share(device, with_rooflines=True):
    """configs[4], one GPU's share, with contacts that bind"""
    # The 18 frames are single wall-clock measurements of 1-5 ms on a host this process shares: the scene is run twice, on two
    # handles (the same deterministic frames: same contacts, same CG budgets), and every frame is given the smaller of its two
    # times (`runs` in the report; a stall of the library's own - a graph instantiated inside a frame - would show in both).
    runs = []
    for rep in range(2):
        if rep:
            g.close()
        g = contact_scene(capi, device)
        g.finalize()
        # one tick to take the first replay of the graph (upload of the executable graph, first touch of the scratch arrays: 8-18
        # ms that are not the solver's), then the scene is put back to its start
        p0, q0, v0 = g.positions.copy(), g.prev_positions.copy(), g.velocities.copy()
        g.tick_async(1)
        g.synchronize()
        g.set_positions(p0); g.set_prev_positions(q0); g.set_velocities(v0)
        run = []
        for _ in range(18):  # the small body lands in frames 0-4 (thousands of contacts bind), then both bodies - w = 1 against m/h^2 = 6944 is
            t0 = time.perf_counter()  # jelly - sag together and the contacts are gone: both regimes are reported
            g.tick_async(1)
            g.synchronize()
            run.append((time.perf_counter() - t0, len(g.tri_collisions), g.pcg_health()["budget"]))
        runs.append(run)
    same = [a[1:] == b[1:] for a, b in zip(*runs)]
    frames = [min(a, b) if eq else b for a, b, eq in zip(runs[0], runs[1], same)]
    binding = [f for f in frames if f[1] > 0]
    quiet = [f for f in frames[6:] if f[1] == 0]
    res, iters, solves = g.pcg_stats()
    ms = sorted(1e3 * f[0] for f in frames[1:])
    out = {"value": len(binding) / max(1e-9, sum(f[0] for f in binding)), "unit": "substeps/s",
           "workload": "BASELINE configs[4], one GPU's share: 25x25x400 beam (250 000 particles) on the floor + an 8x6x30 body "
           "landing on it, PD, strain + volume constraints, 10 iterations, floor and point-triangle contacts (w = 1e4); a host "
           "that synchronises once per frame; value = the frames in which point-triangle contacts bind (contact onset: the CG "
           "budget starts at 32)",
           "frames_with_contacts": len(binding), "contacts_per_frame": [f[1] for f in frames],
           "runs": {"count": 2, "frames_identical_in_both": all(same), "ms_per_frame_first": [round(1e3 * f[0], 3) for f in runs[0]],
                    "ms_per_frame_second": [round(1e3 * f[0], 3) for f in runs[1]],
                    "note": "every frame's time is the smaller of its two runs (wall-clock frames on a shared host)"},
           "cg_budget_per_frame": [f[2] for f in frames], "ms_per_frame": [round(1e3 * f[0], 3) for f in frames],
           # the key of rounds 1-3 keeps its meaning (largest frame over the median of ALL frames after the first); the
           # per-regime measure round 4 introduced has a name of its own
           "max_over_median_frame": ms[-1] / ms[len(ms) // 2],
           "max_over_regime_median_frame": frame_spread([f[0] for f in frames[1:]], [f[1] for f in frames[1:]])[0],
           "first_contact_frame_over_binding_median": (frames[0][0] / sorted(f[0] for f in binding)[len(binding) // 2]) if binding and frames[0][1] > 0 else None,
           "binding_over_quiet_frame": frame_spread([f[0] for f in frames[1:]], [f[1] for f in frames[1:]])[1],
           "max_over_median_all_frames": ms[-1] / ms[len(ms) // 2],
           "frame_spread_note": "max_over_regime_median_frame compares every frame after the first with the median of the frames of its own "
           "regime (contacts binding or not); max_over_median_frame (= max_over_median_all_frames) is the largest frame over the median of "
           "all frames, as in rounds 1-3 - in round 4 the CG budget comes down three frames after the contacts are gone instead of "
           "fourteen, the contact-free frames are 2.4 x faster than in round 3, and that ratio now measures the difference between the "
           "two regimes (binding_over_quiet_frame), not stalls",
           "value_without_tri_contacts": len(quiet) / max(1e-9, sum(f[0] for f in quiet)) if quiet else None,
           "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters, "pcg_health": g.pcg_health(), "failed": g.failed,
           "launches_per_substep": sum(g.launch_counts().values()),
           }
    if with_rooflines:
        out.update(pd_rooflines(g, "contacts", 2))
    g.close()
    return out


def run_config5_rank(device, dims, barrier, dist):
    """BASELINE configs[4] as a multi-rank workload: every rank owns one body of the config-5 pattern (lying on the floor, a
    small body landing on it: floor and point-triangle contacts bind) and runs 18 frames of one tick + one synchronisation;
    value = the frames of all ranks over the slowest rank's time (the same max / sum reduce as the headline)."""
    g = contact_scene(capi, device, dims)
    g.finalize()
    p0, q0, v0 = g.positions.copy(), g.prev_positions.copy(), g.velocities.copy()
    g.tick_async(1)  # first replay of the graph (upload of the executable graph, first touch of scratch arrays), then back to the start
    g.synchronize()
    g.set_positions(p0); g.set_prev_positions(q0); g.set_velocities(v0)
    frames = 18
    barrier()
    t0 = time.perf_counter()
    contacts = 0
    for _ in range(frames):
        g.tick_async(1)
        g.synchronize()
        contacts = max(contacts, len(g.tri_collisions))
    barrier()
    el = time.perf_counter() - t0
    failed = bool(g.failed)
    n = g.count(capi.NODES)
    g.close()
    el, total = aggregate(el, frames, dist)
    return {"value": total / el, "unit": "substeps/s", "frames_per_rank": frames, "particles_per_rank": n, "max_tri_contacts_rank0": contacts,
            "failed_rank0": failed, "workload": "BASELINE configs[4] pattern, one body per rank: %dx%dx%d beam on the floor + an 8x6x30 body "
            "landing on it, PD, strain + volume constraints, 10 iterations, floor and point-triangle contacts; one tick + one "
            "synchronisation per frame, contact onset included" % tuple(dims)}


def run_pd_contacts(device):
    """PD with thousands of contacts: a short beam resting on a long one that lies on the floor"""
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=device)
    g.create_tet_box(25, 25, 160, translation=(0.0, 0.04, 0.0), w=1.0, volume=True, triangles=True)
    g.create_tet_box(25, 25, 40, translation=(0.3, 0.04 + 24 + 0.07, 10.3), w=1.0, volume=True, triangles=True)
    g.finalize()
    # Frames 0-19: the top beam lands and the captured CG budget settles.  Frames 20-35, measured: 29k contacts bind.
    # (Around frame 40 the jelly bodies - w = 1 against m/h^2 = 6944 - bounce apart and re-bind, and after ~100 frames they
    # have sagged through the floor far enough for the reference's own latch, more than 1000 triangles in a grid cell, to
    # end the simulation.)
    frame_loop(g, 20)
    rate = frame_loop(g, 16)
    res, iters, solves = g.pcg_stats()
    out = {"value": rate, "unit": "substeps/s", "workload": "125000 particles: a 25x25x40 beam resting on a 25x25x160 beam on the "
           "floor, PD, 10 iterations, floor + point-triangle contacts binding (w = 1e4 on the diagonal)",
           "tri_contacts_last_substep": len(g.tri_collisions), "pcg_max_rel_residual": res, "pcg_max_iterations_used": iters,
           "pcg_health": g.pcg_health(), "failed": g.failed,
           "roofline_spmv": roofline(g, "pd_spmv", pd_bytes(g)["pd_spmv"], substeps=1, workload="pdcontacts")}
    g.close()
    return out


def run_config4(device):
    """configs[3]: 500k loose particles, node-node collisions + floor, PBD, 4 iterations"""
    p, v = config4_particles()
    g = capi.Solver(scenes.pbd_options(capi, 4), device=device)
    g.addNodes(p)
    g.set_velocities(v)
    g.finalize()
    g.tick_async(2)
    g.synchronize()
    g.collision_stats()
    el = timed_ticks(g, 10, 0, lambda: None)
    pairs, cand = g.collision_stats()
    n = g.count(capi.NODES)
    per_node = 32.0 + 27 * 8.0 + 16.0 * cand / (10 * 4 * n)  # own state + 27 cell headers + 16 B per candidate neighbour
    # The window above (ticks 2-11) is the over-packed block bursting apart: deep dependency orders, two repeated passes, and - in
    # one asynchronous call - the level launches captured for the deepest order seen before it.  A host that synchronises every
    # frame lets the captured launches follow the passes; ten such frames once the burst is over (ticks 12-21):
    t0 = time.perf_counter()
    for _ in range(10):
        g.tick_async(1)
        g.synchronize()
    settled = 10 / (time.perf_counter() - t0)
    _, cand_settled = g.collision_stats()
    per_node = 32.0 + 27 * 8.0 + 16.0 * cand_settled / (10 * 4 * n)  # (the brackets below are taken in this state)
    # The grid rebuild: SURVEY 8d's 92 B (key generation 20 + four sort passes of 16 + cell index 8) are the bytes of ONE (cell, node)
    # entry; a node is entered into every cell of its range (NodeCompRange, Solver.cpp:877-901: ceil(fract(min) + 2R) cells per axis -
    # 2 x 2 x 2 for this scene's radius 0.5, padding 0.5 and grid spacing 2).  Priced per entry:
    pos_now, rad_now = g.positions.astype(np.float32), g.radii.astype(np.float32)
    Rn = ((rad_now + np.float32(0.5)) / np.float32(2.0))[:, None]
    mn = pos_now / np.float32(2.0) - Rn
    entries_per_node = float(np.prod(np.ceil((mn - np.floor(mn)) + 2 * Rn), axis=1).mean())
    out = {"value": 10 / el, "unit": "substeps/s", "settled_value": settled,
           "settled_note": "ticks 12-21, one tick and one synchronisation per frame (the burst of the over-packed block is over, the "
                           "captured level launches have followed the passes down)",
           "workload": "BASELINE configs[3]: 50x100x100 loose particles (r 0.5, spacing "
           "0.9, jitter 0.05), PBD, 4 iterations, grid rebuild + node-node resolve + floor every iteration, parallel "
           "collision order", "resolved_pairs_per_substep": pairs / 10, "candidates_per_node_per_iteration": cand / (40 * n),
           "failed": g.failed, "launches_per_substep": sum(g.launch_counts().values()),
           "roofline": roofline(g, "collide", per_node, substeps=1, workload="config4", note="one bracket = the resolve pass of one "
                                "iteration, taken after tick 22 (settled state); bytes per node = 32 + 27 x 8 + 16 x candidates the "
                                "reference's loop looks at in that state"),
           "grid_entries_per_node": entries_per_node,
           "roofline_grid_build": roofline(g, "hash", 92.0 * entries_per_node, substeps=1, workload="config4", note="one bracket = one grid "
                                           "rebuild: range, prefix sum, emit, radix sort passes, cell index.  Bytes = SURVEY 8d's 92 B per "
                                           "(cell, node) ENTRY x %.2f entries per node (a node is entered into every cell of its range; "
                                           "rounds 1-5 priced one entry per node, which made this fraction unusable)" % entries_per_node)}
    # The same scene in the REFERENCE's node-node order (ascending node index, range from the live position: Solver.cpp:85-130), by
    # dependency levels of turns (round 5), from the settled state the run above has reached: frames of one tick + one synchronisation
    state = (g.positions, g.velocities)
    g.close()
    try:
        r = capi.Solver(scenes.pbd_options(capi, 4), device=device)
        r.addNodes(state[0])
        r.set_velocities(state[1])
        r.set_flag(capi.FLAG_COLLISION_ORDER, capi.COLLISION_ORDER_REFERENCE)
        r.finalize()
        for _ in range(3):  # (the captured level launches follow the passes at the synchronisations)
            r.tick_async(1)
            r.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r.tick_async(1)
            r.synchronize()
        out["reference_order"] = {"value": 5 / (time.perf_counter() - t0), "unit": "substeps/s", "collision_health": r.collision_health(),
                                  "passes_left_to_the_sequential_loop": r.collision_fallbacks, "failed": r.failed,
                                  "launches_per_substep": sum(r.launch_counts().values()),
                                  "note": "the reference's own visiting order, bit-identical to the oracle's plain loop (tests/test_collisions_gpu.py"
                                          "::test_config4_l500k_one_tick_reference_order_settled); settled state (ticks 23-30); until round 4 this "
                                          "order was one sequential chain on one wavefront: 0.027 substeps/s"}
        r.close()
    except Exception as e:  # noqa: BLE001
        out["reference_order"] = {"error": str(e)}
    return out


def run_config2_default_tick(device, dims, steps, with_exact):
    """The reference's DEFAULT PBD tick: tickPBD runs the hash rebuild and the node-node pass in every iteration unconditionally
    (Src/Solver.cpp:81-130).  Two 100k-particle beams: BASELINE configs[1] itself (createTetBox nodes have radius 0.475) - there the
    PBD tetrahedral projection (quirk Q2) drags the whole body into a few grid cells within the first tick, which is a pile-up no
    grid handles (the reference would visit ~10^10 pairs per iteration; this build latches its "more than 2048 nodes in a cell"
    failure) - and the same lattice with the distance constraints alone (createBox pattern, radius 0.5 = touching spheres).  That
    scene is not stable under the reference's semantics either (one-sided distance projections, quirk Q1, against the
    collision pass: the oracle's plain loops blow an 8x8x30 box up from 7 to 11 wide in 12 ticks, and so does every device order),
    so the figure is for its FIRST FIVE ticks, while the lattice is still a lattice.  LAYERED + pair order, and (with_exact)
    schedule EXACT = the reference's order end to end (one tick: its node-node loop is one sequential chain)."""
    out = {}
    for scene, tets in (("config2", True), ("box_100k_distance_only", False)):
        res = out[scene] = {"workload": "%dx%dx%d lattice, %s, PBD, %d iterations, node-node pass ON (what Solver::tickPBD runs by default)"
                            % (dims + ("distance + tet-strain constraints (BASELINE configs[1])" if tets else "distance constraints only (createBox pattern)", ITERATIONS))}
        for name, sched, n in (("layered", capi.SCHEDULE_LAYERED, min(steps, 5)), ("exact", capi.SCHEDULE_EXACT, 1)):
            if name == "exact" and (not with_exact or tets):
                continue
            # (a pile the parallel orders cannot run is left to the reference's sequential loop since round 5 - unless the pass would cost
            # more candidate tests than PIES_FALLBACK_VISITS: the collapsing config-2 body reaches 1e10 per iteration, hours per tick in the
            # reference as well; the budget is kept small here so that the section ends)
            capi.set_tuning("PIES_FALLBACK_VISITS", "5000000")
            try:
                g = capi.Solver(scenes.pbd_options(capi, ITERATIONS), device=device)
                scenes.build_beam(g, dims, tets=tets)
                scenes.perturb(g, 1234, 0.05)
                g.set_flag(capi.FLAG_NODE_COLLISIONS, 1)
                g.set_schedule(sched)
                g.finalize()
                el = timed_ticks(g, n, 0, lambda: None)
                failed = bool(g.failed)
                res[name] = {"value": None if failed else n / el, "unit": "substeps/s", "steps": n, "launches_per_substep": sum(g.launch_counts().values()),
                             "failed": failed, "error": g.last_error() if failed else None, "collision_health": g.collision_health(),
                             "passes_left_to_the_sequential_loop": g.collision_fallbacks}
                g.close()
            finally:
                capi.set_tuning("PIES_FALLBACK_VISITS", None)
    return out


def scale_profiles(device, with_coloured=True):
    """The same kernels at 1M particles (100x100x100), where a launch is long enough for HBM rather than the kernel boundary
    to bound it: whole-substep throughput and in-situ rooflines."""
    out = {}
    for name, sched in (("pbd_1m", capi.SCHEDULE_LAYERED), ("pbd_1m_coloured", capi.SCHEDULE_COLOURED)):
        if sched == capi.SCHEDULE_COLOURED and not with_coloured:
            continue
        log(name)
        g = build_scene(capi, scenes.L1M, 99, schedule=sched, device=device)
        g.finalize()
        el = timed_ticks(g, 3, 1, lambda: None)
        out[name] = {"substeps_per_sec": 3 / el, "projections_per_sec": 3 / el * scenes.projections_per_substep(g, capi, ITERATIONS),
                     "launches_per_substep": sum(g.launch_counts().values())}
        if sched == capi.SCHEDULE_LAYERED:
            out[name]["roofline"] = roofline(g, "layer", 1, substeps=1, workload="pbd1m")
        else:
            out[name]["roofline"] = roofline(g, "tet", BYTES["tet"], substeps=1, workload="none")
            out[name]["roofline_distance"] = roofline(g, "distance", BYTES["distance"], substeps=1, workload="none")
        g.close()
    log("pd_1m")
    g = pd_beam(scenes.L1M, device, settle=12)
    el = timed_ticks(g, 5, 1, lambda: None)
    out["pd_1m"] = {"substeps_per_sec": 5 / el, "pcg_stats": g.pcg_stats(), "pcg_health": g.pcg_health(), "tiles": g.count(capi.PD_TILES),
                    "tile_records_per_node": g.count(capi.PD_TILE_RECORDS) / max(1, g.count(capi.NODES)),
                    "launches_per_substep": sum(g.launch_counts().values())}
    out["pd_1m"].update(pd_rooflines(g, "pd1m", 1))
    if not g.count(capi.PD_CG_SINGLE):
        out["pd_1m"]["roofline_cg_update"] = roofline(g, "pd_cg_update", pd_bytes(g)["pd_cg_update"], substeps=1, workload="pd1m")
    g.close()
    # the same body with the row dictionary off: the lattice streams its matrix like an unstructured mesh does (windowed SELL)
    log("pd_1m_streamed")
    capi.set_tuning("PIES_PD_ROW_DICT", "0")
    try:
        g = pd_beam(scenes.L1M, device, settle=12)
        el = timed_ticks(g, 5, 1, lambda: None)
        out["pd_1m_streamed"] = {"substeps_per_sec": 5 / el, "pcg_health": g.pcg_health(), "row_stencils": g.count(capi.ROW_STENCILS),
                                 "launches_per_substep": sum(g.launch_counts().values()),
                                 "note": "PIES_PD_ROW_DICT=0: no row shares its stencil with another, the CG iterations stream the matrix"}
        B = pd_bytes(g)
        sp = roofline(g, "pd_spmv", B["pd_spmv"], substeps=1, workload="pd1m_streamed", note=(
            "k_cg1_iter with the matrix streamed (windowed SELL: value + 16-bit window slot per entry, a chunk's columns staged in LDS "
            "once): SURVEY 8d's 8 nnz + 148 N bytes per launch; frac_required_bytes prices the launch by what this form has to move"))
        if sp:
            sp["frac_spmv_bytes_only"] = sp["frac"] * B["pd_spmv_only"] / B["pd_spmv"]
            sp["frac_required_bytes"] = sp["frac"] * B["pd_spmv_required"] / B["pd_spmv"]
            sp["required_bytes_per_row"] = B["pd_spmv_required"]
        out["pd_1m_streamed"]["roofline_spmv"] = sp
        g.close()
    finally:
        capi.set_tuning("PIES_PD_ROW_DICT", None)
    return out


COMPACT_LIMIT = 4096  # bytes: the driver parses the final stdout line; round 2's 30 KB line was dropped


def _r(x, digits=4):
    """Round floats to `digits` significant figures for the compact line."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d and k in d}


def compact_line(full):
    """The ONE stdout line the driver parses: the contract keys, `roofline`, `cpu_baseline` and a dozen scalars.  Everything
    else (prose, per-frame arrays, per-class latencies, order deviation) lives in bench_full.json."""
    c = {k: _r(full[k], 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data") if k in full}
    c["config"] = dict(full["config"])
    c["projections_per_sec"] = _r(full["projections_per_sec"], 6)
    if full.get("roofline"):
        c["roofline"] = _pick(full["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                 "bytes_per_launch", "rocprofv3_avg_us", "frac_in_situ", "valu_issue_frac", "overhead_clamped"))
    else:
        c["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "host_cpus", "projections_per_sec"))
        if cb.get("all_cores"):
            c["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
    s = {}

    def put(name, *path):
        d = full
        for p in path:
            d = d.get(p) if isinstance(d, dict) else None
            if d is None:
                return
        s[name] = _r(d)
    put("exact_order", "exact_order", "value")  # the reference's own visiting order on the same scene (the headline's LAYERED order is a re-ordered sweep)
    put("unstructured_value", "other_configs", "unstructured_config2", "layered", "value")  # BASELINE says "tetgen beam": the Delaunay stand-in
    put("unstructured_pd_value", "other_configs", "unstructured_config2", "pd", "value")
    put("unstructured_pd_frac_pcg_iter", "other_configs", "unstructured_config2", "pd", "roofline_spmv", "frac")
    put("coloured", "coloured_schedule", "value")
    put("pies_tick", "tick_inclusive", "pies_tick_substeps_per_sec")
    put("config2_collisions_on", "config2_default_tick", "config2", "layered", "value")
    put("config2_collisions_on_failed", "config2_default_tick", "config2", "layered", "failed")
    put("box100k_collisions_on", "config2_default_tick", "box_100k_distance_only", "layered", "value")
    put("box100k_collisions_on_exact", "config2_default_tick", "box_100k_distance_only", "exact", "value")
    put("config3_value", "other_configs", "pd_config3", "value")
    put("config3_frac_local", "other_configs", "pd_config3", "roofline", "frac")
    put("config3_valu_frac_local", "other_configs", "pd_config3", "roofline", "valu_frac")
    put("pd_1m_valu_frac_local", "scale_1m", "pd_1m", "roofline", "valu_frac")
    put("config3_frac_spmv", "other_configs", "pd_config3", "roofline_spmv", "frac_spmv_bytes_only")
    put("config3_frac_pcg_iter", "other_configs", "pd_config3", "roofline_spmv", "frac")
    put("config4_value", "other_configs", "collisions_config4", "value")
    put("config4_settled", "other_configs", "collisions_config4", "settled_value")
    put("config4_frac_resolve", "other_configs", "collisions_config4", "roofline", "frac")
    put("config4_frac_grid", "other_configs", "collisions_config4", "roofline_grid_build", "frac")
    put("config4_reference_order", "other_configs", "collisions_config4", "reference_order", "value")
    put("config5_share_value", "other_configs", "pd_config5_per_gpu", "value")
    put("config5_max_over_median_frame", "other_configs", "pd_config5_per_gpu", "max_over_median_frame")
    put("config5_max_over_regime_median_frame", "other_configs", "pd_config5_per_gpu", "max_over_regime_median_frame")
    put("config5_binding_over_quiet_frame", "other_configs", "pd_config5_per_gpu", "binding_over_quiet_frame")
    put("config5_quiet_value", "other_configs", "pd_config5_per_gpu", "value_without_tri_contacts")
    put("pd_contacts_value", "other_configs", "pd_contacts", "value")
    put("pbd_1m_value", "scale_1m", "pbd_1m", "substeps_per_sec")
    put("pbd_1m_frac", "scale_1m", "pbd_1m", "roofline", "frac")
    put("pbd_1m_frac_rocprofv3", "scale_1m", "pbd_1m", "roofline", "frac_rocprofv3")
    put("pd_1m_value", "scale_1m", "pd_1m", "substeps_per_sec")
    put("pd_1m_frac_local", "scale_1m", "pd_1m", "roofline", "frac")
    put("pd_1m_frac_spmv", "scale_1m", "pd_1m", "roofline_spmv", "frac_spmv_bytes_only")
    put("pd_1m_frac_pcg_iter", "scale_1m", "pd_1m", "roofline_spmv", "frac")
    put("pd_1m_frac_rhs", "scale_1m", "pd_1m", "roofline_rhs", "frac")
    put("pd_1m_frac_pcg_iter_required", "scale_1m", "pd_1m", "roofline_spmv", "frac_required_bytes")
    put("pd_1m_streamed_value", "scale_1m", "pd_1m_streamed", "substeps_per_sec")
    put("pd_1m_streamed_frac_pcg_iter", "scale_1m", "pd_1m_streamed", "roofline_spmv", "frac")
    put("pd_1m_streamed_frac_required", "scale_1m", "pd_1m_streamed", "roofline_spmv", "frac_required_bytes")
    put("pd_1m_streamed_frac_rocprofv3", "scale_1m", "pd_1m_streamed", "roofline_spmv", "frac_rocprofv3")
    put("config5_value", "config5_all_ranks", "value")
    put("value_export_inclusive", "tick_inclusive", "async_export_substeps_per_sec")
    c.update(s)
    if full.get("errors"):
        c["errors"] = len(full["errors"])
    if full.get("skipped"):
        c["skipped"] = len(full["skipped"])
    c["full_report"] = "bench_full.json"
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:  # never let extras cost the record: drop the scalars, then the sample prose
        for k in list(s):
            c.pop(k, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT and c.get("cpu_baseline"):
        c["cpu_baseline"]["sample"] = c["cpu_baseline"]["sample"][:160]
        c["config"]["workload"] = c["config"]["workload"][:160]
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def _finite(x):
    """json.dumps(allow_nan=False) refuses nan/inf: the full report maps them to None instead of failing the run."""
    if isinstance(x, float):
        return x if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, np.generic):
        return _finite(x.item())
    return x


def write_full(full):
    text = json.dumps(_finite(full), allow_nan=False, indent=1)
    paths = [os.path.join(ROOT, "bench_full.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(text)
        except OSError as e:
            log("could not write %s: %s" % (p, e))
    print(text, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dims", type=int, nargs=3, default=list(scenes.L100K))
    ap.add_argument("--schedule", choices=["layered", "coloured", "exact"], default="layered")
    ap.add_argument("--quick", action="store_true", help="headline + roofline + a one-tick CPU baseline only")
    ap.add_argument("--full", action="store_true", help="also: order deviation, unstructured beam, config 5 share, contact scene, 1M "
                    "particles, per-class latencies, CPU baselines of configs 3 and 4, the reference-order default tick (minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the other schedules' throughput")
    ap.add_argument("--no-roofline", action="store_true", help="skip the in-situ timing pass (roofline = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip BASELINE configs 3 and 4")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of CPU work per baseline sample")
    ap.add_argument("--config5-dims", type=int, nargs=3, default=list(scenes.L250K), help="the large body of the config-5 region (tests shrink it)")
    ap.add_argument("--time-budget", type=float, default=80.0, help="seconds after which the default run starts no further optional section "
                    "(they are named in `skipped`; --full ignores it)")
    args = ap.parse_args()
    if args.quick:
        args.no_exact = args.no_extras = True
        args.full = False
        args.cpu_budget = min(args.cpu_budget, 1.0)

    rank, local_rank, world = dist_env()
    dist = None
    # PIES_BENCH_FORCE_DIST=1: the N > 1 branch with a world of ONE rank (tests/test_bench_distributed.py): process-group init over RCCL,
    # the 4-byte all-reduce barrier and aggregate() on device tensors, on a one-GPU box
    if world > 1 or os.environ.get("PIES_BENCH_FORCE_DIST") == "1":
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
    assert np.isfinite(g.positions).all() and not g.failed

    result = None
    if rank == 0:
        value = total_substeps / elapsed
        lc = g.launch_counts()
        dom = "layer" if lc.get("layer") else "wave" if lc.get("wave") else "tet"
        per_unit = 1
        if dom == "tet":
            per_unit = BYTES["tet"]
        elif dom == "wave":  # schedule exact: a launch mixes the kinds of one dependency level
            items = ITERATIONS * (g.count(capi.DISTANCE) + g.count(capi.TET) + g.count(capi.POSITION) + g.count(capi.BEND) + g.count(capi.NODES))
            per_unit = ITERATIONS * (BYTES["distance"] * g.count(capi.DISTANCE) + BYTES["tet"] * g.count(capi.TET) + BYTES["position"]
                                     * g.count(capi.POSITION) + BYTES["bend"] * g.count(capi.BEND) + BYTES["floor"] * g.count(capi.NODES)) / items
        result = {
            "metric": "substeps/sec @100k particles (PBD distance+tet-strain, 20 iterations)",
            "value": value, "unit": "substeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %dx%dx%d lattice beam, %d particles, %d distance + %d tet-strain "
                                   "constraints, PBD, %d iterations, 1 substep/tick, collisions off, one body per GPU"
                                   % (dims + (g.count(capi.NODES), g.count(capi.DISTANCE), g.count(capi.TET), ITERATIONS)),
                       "schedule": args.schedule, "parallelism": "replicas x%d" % world, "launches_per_substep": sum(lc.values()),
                       "value_is": "state resident in HBM across ticks (pies_tick_async); value_export_inclusive = the same ticks with "
                                   "every frame's positions delivered to the host (asynchronous export), pies_tick = the synchronous "
                                   "Pies::Solver::tick"},
            "projections_per_sec": value * proj,
            "roofline": None if args.no_roofline else roofline(g, dom, per_unit, whole_graph=lc.get(dom, 0) == sum(lc.values()),
                                                               note="algorithmic bytes = the projections and per-node steps "
                                                               "a launch executes (160 B per tet, 52 per distance, 44 per position constraint, 136 per "
                                                               "bend, 48 / 20 / 40 per node for predict / floor / velocity), tallied by the library"),
            "errors": [],
        }
        if result["roofline"]:
            result["roofline"]["timed_region_us_per_launch"] = 1e6 * (elapsed / args.steps) / max(1, sum(lc.values()))
            if dom == "layer" and dims == tuple(scenes.L100K):
                vi = valu_issue_of_layer(result["roofline"].get("rocprofv3_avg_us") or result["roofline"]["avg_launch_us"])
                if vi:
                    result["roofline"]["valu_issue"] = vi
                    result["roofline"]["valu_issue_frac"] = vi["frac"]
        # whole-substep algorithmic traffic over wall time (includes launch gaps)
        per_substep_bytes = (BYTES["predict"] + BYTES["velocity"] + ITERATIONS * BYTES["floor"]) * g.count(capi.NODES) + ITERATIONS * (
            BYTES["distance"] * g.count(capi.DISTANCE) + BYTES["tet"] * g.count(capi.TET) + BYTES["position"] * g.count(capi.POSITION)
            + BYTES["bend"] * g.count(capi.BEND))
        result["substep_algorithmic_GBs_per_gpu"] = per_substep_bytes * (value / world) / 1e9
    g.close()

    # BASELINE configs[4] on every rank (N > 1: the second timed region; N = 1 runs the same scene as a section below)
    c5 = None
    if world > 1:
        log("config 5 region (one contact scene per rank)")
        c5 = run_config5_rank(device_index, tuple(args.config5_dims), barrier, dist)
    if rank == 0 and c5 is not None:
        result["config5_all_ranks"] = c5

    if rank == 0:
        def section(name, fn, *a, **kw):
            """An extra must never cost the record: a failing section is logged and named in `errors`."""
            log(name)
            try:
                return fn(*a, **kw)
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                result["errors"].append("%s: %s" % (name, e))
                return None

        one = world == 1
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = section("CPU baseline (config 2)", cpu_baseline, dims, args.cpu_budget,
                                             with_all_cores=one and not args.quick)
        if one and not args.quick:
            result["tick_inclusive"] = section("tick-inclusive figures", tick_inclusive, dims, device_index, sched, max(10, min(args.steps, 50)))
        if one and not args.no_exact and args.schedule == "layered":
            def coloured():
                c = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_COLOURED, device=device_index)
                c.finalize()
                steps = max(2, min(args.steps, 50))
                el = timed_ticks(c, steps, 2, lambda: None)
                out = {"value": steps * substeps_per_tick / el, "unit": "substeps/s",
                       "launches_per_substep": sum(c.launch_counts().values()), "steps": steps,
                       "roofline": roofline(c, "tet", BYTES["tet"], workload="none"),
                       "note": "schedule COLOURED: one launch per colour class (24 tet + 9 distance colours per iteration)"}
                if args.full:
                    out["isolated_replay_latencies"] = replay_latencies(c)
                c.close()
                return out
            result["coloured_schedule"] = section("coloured schedule", coloured)
        if one and not args.no_exact and args.schedule != "exact":
            def exact():
                e = build_scene(capi, dims, 1234, schedule=capi.SCHEDULE_EXACT, device=device_index)
                e.finalize()
                steps = max(2, min(args.steps, 10))
                el = timed_ticks(e, steps, 1, lambda: None)
                out = {"value": steps * substeps_per_tick / el, "unit": "substeps/s",
                       "launches_per_substep": sum(e.launch_counts().values()), "steps": steps,
                       "note": "schedule EXACT: the reference's order (containers swept sequentially in insertion order); the device "
                               "result is bit-identical to the ORACLE's container-order sweep (tests/test_pbd_parity_gpu.py) - the "
                               "oracle restates the reference with its own 3x3 SVD, so against Eigen's JacobiSVD this is a rounding-"
                               "level tolerance, not bit equality; one launch per level of the whole-substep dependency DAG"}
                e.close()
                return out
            result["exact_order"] = section("exact schedule", exact)
        if one and not args.no_extras:
            result["skipped"] = []

            def optional(name, fn, *a, **kw):
                """Sections of the default run beyond configs 2-4: started only while the run is inside its time budget."""
                if not args.full and time.perf_counter() - T_START > args.time_budget:
                    result["skipped"].append(name)
                    log("skipped (time budget): " + name)
                    return None
                return section(name, fn, *a, **kw)

            oc = result["other_configs"] = {}
            oc["pd_config3"] = section("config 3 (PD)", run_config3, device_index, args.full)
            oc["collisions_config4"] = section("config 4 (500k particles, node-node collisions)", run_config4, device_index)
            result["scale_1m"] = optional("1M-particle measurements", scale_profiles, device_index, args.full)
            oc["pd_config5_per_gpu"] = optional("config 5 share (250k particles, PD, binding contacts)", run_config5_share, device_index, args.full)
            oc["unstructured_config2"] = optional("unstructured beam (PBD and PD)", run_unstructured, device_index, args.full)
            result["config2_default_tick"] = optional("config 2 with the node-node pass on", run_config2_default_tick, device_index, dims,
                                                      max(2, min(args.steps, 20)), args.full)
            if args.full:
                oc["pd_contacts"] = section("PD contact scene", run_pd_contacts, device_index)
                result["order_deviation"] = section("order deviation", order_deviation, device_index)
                if not args.no_cpu_baseline:
                    if oc.get("pd_config3"):
                        oc["pd_config3"]["cpu_baseline"] = section("CPU baseline (config 3)", cpu_baseline_pd, scenes.L100K, args.cpu_budget)
                    if oc.get("collisions_config4"):
                        oc["collisions_config4"]["cpu_baseline"] = section("CPU baseline (config 4)", cpu_baseline_collisions, args.cpu_budget)
                    if result.get("config2_default_tick"):
                        result["config2_default_tick"]["box_100k_distance_only"]["cpu_baseline"] = section("CPU baseline (config 2, node-node pass on)",
                                                                                 cpu_baseline_default_tick, dims, args.cpu_budget)
        log("done")
        write_full(result)
        sys.stderr.flush()
        print(compact_line(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
