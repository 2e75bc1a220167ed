"""The one stdout line of bench.py: the driver parses it, so it must stay small (round 2's 30 KB line was dropped), strict
JSON (no NaN) and carry the contract keys whatever the extras hold.  CPU only: compact_line() is fed a synthetic report."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "projections_per_sec", "roofline", "cpu_baseline")


def fake_report(prose=200):
    rl = {"bound": "hbm", "kernel": "k_layer", "achieved": 1690.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.2112654,
          "traffic": 22.5e6, "avg_launch_us": 35.2, "bytes_per_launch": 59.75e6, "rocprofv3_avg_us": 36.0, "method": "x" * prose,
          "note": "y" * prose, "traffic_source": "profiles/..."}
    return {
        "metric": "substeps/sec @100k particles (PBD distance+tet-strain, 20 iterations)", "value": 692.123456789, "unit": "substeps/s",
        "n_gpus": 1, "steps": 200, "warmup": 20, "ms_per_step": 1.4451234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 20x20x250 lattice beam, 100000 particles, 649156 distance + 539334 tet-strain "
                               "constraints, PBD, 20 iterations, 1 substep/tick, collisions off, one body per GPU", "schedule": "layered",
                   "parallelism": "replicas x1", "launches_per_substep": 41},
        "projections_per_sec": 1.64e10, "roofline": rl,
        "cpu_baseline": {"value": 0.43, "unit": "substeps/s", "cores": 1, "kind": "port", "sample": "s" * prose, "host_cpus": 128,
                         "projections_per_sec": 1e7, "all_cores": {"value": 4.4, "cores": 16, "sample": "z" * prose}},
        "exact_order": {"value": 60.0, "note": "n" * prose}, "coloured_schedule": {"value": 331.0, "isolated_replay_latencies": {"a": 1}},
        "tick_inclusive": {"pies_tick_substeps_per_sec": 615.0, "async_export_substeps_per_sec": 673.0},
        "other_configs": {"pd_config3": {"value": 1330.0, "roofline": dict(rl), "roofline_spmv": dict(rl)},
                          "collisions_config4": {"value": 26.0, "roofline": dict(rl), "roofline_grid_build": dict(rl)},
                          "pd_config5_per_gpu": {"value": 246.0, "ms_per_frame": list(range(1000)), "max_over_median_frame": 1.4}},
        "order_deviation": {"blob": "d" * 20000}, "errors": [],
    }


def test_compact_line_is_small_strict_json_with_the_contract_keys():
    line = bench.compact_line(fake_report())
    assert len(line) <= 4096 and "\n" not in line
    r = json.loads(line)
    for k in REQUIRED:
        assert k in r, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(r["cpu_baseline"])
    assert "model" not in r["config"] and r["config"]["workload"].startswith("BASELINE configs[1]")
    assert r["exact_order"] == 60.0 and r["config3_value"] == 1330.0 and r["config4_frac_grid"] == pytest.approx(0.2113)
    assert "order_deviation" not in r and "other_configs" not in r


def test_compact_line_sheds_extras_before_it_grows():
    line = bench.compact_line(fake_report(prose=5000))
    assert len(line) <= 4096
    r = json.loads(line)
    for k in REQUIRED:
        assert k in r, k


def test_compact_line_refuses_nan():
    rep = fake_report()
    rep["value"] = float("nan")
    with pytest.raises(ValueError):
        bench.compact_line(rep)


def test_full_report_maps_non_finite_to_null():
    assert bench._finite({"a": [float("inf"), 1.0], "b": float("nan")}) == {"a": [None, 1.0], "b": None}


def test_every_name_bench_uses_is_defined():
    """bench.py's optional sections only run on the GPU box and behind --full: a name that a refactoring lost must not wait
    for that run to be noticed (round 3 lost scale_profiles that way).  Static check: every global name loaded anywhere in the
    module is defined at module level, imported or a builtin."""
    import ast
    import builtins
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    defined = set(dir(builtins)) | {"__file__", "__name__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defined.add(node.name)
            for a in node.args.args + node.args.kwonlyargs if isinstance(node, ast.FunctionDef) else []:
                defined.add(a.arg)
            if isinstance(node, ast.FunctionDef):
                if node.args.vararg:
                    defined.add(node.args.vararg.arg)
                if node.args.kwarg:
                    defined.add(node.args.kwarg.arg)
        elif isinstance(node, ast.Lambda):
            for a in node.args.args:
                defined.add(a.arg)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                defined.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            defined.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            defined.add(node.name)
        elif isinstance(node, ast.comprehension):
            for n in ast.walk(node.target):
                if isinstance(n, ast.Name):
                    defined.add(n.id)
    used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    assert not (used - defined), sorted(used - defined)
