#!/usr/bin/env python3
"""Where does a launch of the one-launch CG iteration (k_cg1_iter) spend its time?  In-kernel time stamps (s_memtime) of the
diagnostic build (python -m pies_amd.build --exp), lane 0 of every workgroup; every captured launch runs as a working iteration
(PIES_PCG_NEVER_EXIT).  usage: python tools/cg_timeline.py [unstructured|lattice] [NAME=VALUE ...] > profiles/r06_cg_timeline.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PIES_LIB"] = os.path.join(ROOT, "pies_amd", "lib", "libpies_hip_exp.so")
for p in (ROOT, os.path.join(ROOT, "benchlib")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import scenes  # noqa: E402
from pies_amd import capi  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "unstructured"
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    capi.set_tuning(k, v)
capi.set_tuning("PIES_PCG_NEVER_EXIT", "1")
capi.set_tuning("PIES_PCG_BUDGET", "3")  # (fixed: a solve that never exits would drive the adaptive budget to its ceiling)
if what == "unstructured":
    g = capi.Solver(capi.Options(solver=capi.PD, iterations=10), device=0)
    scenes.build_unstructured_pd(g, scenes.delaunay_beam(scenes.L100K))
else:
    g = bench.pd_beam(scenes.L100K, 0)
g.finalize()
for _ in range(10):
    g.tick_async(1)
    g.synchronize()
SLOTS, BLOCKS, NST = 8, 4096, 32
buf = torch.zeros(SLOTS * BLOCKS * NST, dtype=torch.int64, device="cuda:0")
L = capi.load()
L.pies_exp_cg_stamps.argtypes = [ctypes.c_void_p]
assert L.pies_exp_cg_stamps(buf.data_ptr()) == 0
g.tick_async(1)
g.synchronize()
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(SLOTS, BLOCKS, NST)
assert not g.failed, g.last_error()
assert L.pies_exp_cg_stamps(None) == 0
print("%s 100k beam, PD; window entries %d, halo per row %.2f, rows per chunk %s; stamps of the substep's last solve" % (
    what, g.count(capi.PD_WINDOW_ENTRIES), g.count(capi.PD_WINDOW_HALO) / max(1, g.count(capi.NODES)), os.environ.get("PIES_CG_CHUNK_ROWS", "default")))
for s in range(SLOTS):
    blocks = np.nonzero(st[s, :, 1])[0]
    if not len(blocks):
        continue
    rows = st[s, blocks]
    n = int((rows[0, :31] != 0).sum())
    real0, real1 = rows[:, 0], rows[:, 31]
    t = rows[:, 1:n].astype(np.float64)
    dur = (real1 - real0) * 10.0  # ns (100 MHz)
    clock = np.median((t[:, -1] - t[:, 0]) / np.maximum(dur, 1.0))
    d = np.diff(t, axis=1) / clock
    print("iteration slot %d: %d workgroups, %d stamps, clock %.2f GHz; launch span %.2f us, workgroup lifetime median %.2f max %.2f us, start skew %.2f us" % (
        s, len(blocks), n, clock, (real1.max() - real0.min()) * 10.0 / 1e3, np.median(dur) / 1e3, dur.max() / 1e3, (real0.max() - real0.min()) * 10.0 / 1e3))
    print("   intervals (ns), median over workgroups: " + " ".join("%.0f" % v for v in np.median(d, axis=0)))
    print("   intervals (ns), 90th percentile       : " + " ".join("%.0f" % v for v in np.quantile(d, 0.9, axis=0)))
print("stamps: entry | scalars (partials re-reduced) | per chunk: own rows, halo, barrier, [per slice: sum, row's turn], barrier | partial sums written")
g.close()
