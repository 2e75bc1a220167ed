"""The N>1 path of bench.py on CPU: world_size 2 over gloo.  Bodies are independent (one per rank, SURVEY
8e "replicas only"): the only communication is the timing record -- max over ranks of the elapsed time and
the sum of substeps -- which is what this test exercises; no solver runs here."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert bench.dist_env() == (rank, rank, world)
    elapsed, units = bench.aggregate(1.0 + 0.5 * rank, 100 * (rank + 1), dist)  # rank 1 is the slow one
    q.put((rank, elapsed, units))
    dist.destroy_process_group()


def test_aggregate_is_max_time_and_summed_units_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, units in res:
        assert elapsed == pytest.approx(1.5) and units == pytest.approx(300.0)
    # whole-job value = all ranks' substeps / slowest rank's time
    assert res[0][2] / res[0][1] == pytest.approx(200.0)


def test_single_process_aggregate_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(2.5, 40, None) == (2.5, 40)


@pytest.mark.gpu
def test_two_rank_bench_end_to_end_on_one_gpu():
    """main()'s N > 1 branch as the driver launches it - one fresh process per rank through torch.distributed.run,
    process-group init, the all-reduce barrier around the timed region, max time / summed substeps, rank 0's JSON line -
    rehearsed on a one-GPU box: both ranks share the card (device = LOCAL_RANK modulo the device count) and the timing
    record travels over gloo (PIES_BENCH_BACKEND; the driver's 8-GPU run uses RCCL)."""
    import json
    import subprocess
    env = dict(os.environ, PIES_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--dims", "10", "10", "60", "--config5-dims", "10", "10", "60", "--quick"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    assert len(lines[0]) <= 4096 and out.stdout.rstrip().endswith(lines[0])  # small, and the LAST thing on stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["config"]["parallelism"] == "replicas x2"
    # value = the substeps of BOTH ranks over the slower rank's time
    assert r["value"] * r["ms_per_step"] * r["steps"] / 1e3 == pytest.approx(2 * 5, rel=1e-4)  # (the line carries 6 significant digits)
    assert r["roofline"]["frac"] > 0 and r["cpu_baseline"]["value"] > 0
    # the second timed region: BASELINE configs[4]'s pattern (PD + contacts), one body per rank, the same max / sum reduce
    assert r["config5_value"] > 0


@pytest.mark.gpu
def test_rccl_branch_runs_on_one_gpu():
    """The RCCL branch itself - init_process_group("nccl"), the 4-byte all-reduce barrier on a device tensor, aggregate() on device
    tensors - executed on real hardware: a world of ONE rank started the way the driver starts N (a fresh process through
    torch.distributed.run; nothing re-executes after the GPU is initialised), with PIES_BENCH_FORCE_DIST=1 taking main() down the
    N > 1 path.  (The two-rank test above has to carry its record over gloo: two ranks cannot share one card under RCCL.)"""
    import json
    import subprocess
    env = dict(os.environ, PIES_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PIES_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--dims", "10", "10", "60", "--quick"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["value"] > 0
    assert r["value"] * r["ms_per_step"] * r["steps"] / 1e3 == pytest.approx(5, rel=1e-4)
