#!/usr/bin/env python3
"""Where does a k_layer launch spend its time?  In-kernel time stamps (s_memtime) of the diagnostic build
(python -m pies_amd.build --exp; PIES_LIB selects it), BASELINE config 2 in the state bench.py times (after 25 ticks).
One stamp after the tile's node records are in LDS, one after every colour's barrier, one at the end; lane 0 of wavefront 0
of every tile.  usage: python tools/layer_timeline.py [ticks_before] > profiles/r06_layer_timeline.txt"""
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

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 25
g = bench.build_scene(capi, scenes.L100K, 1234, schedule=capi.SCHEDULE_LAYERED, device=0)
g.finalize()
g.tick(ticks)
SLOTS, TILES, NST = 64, 4096, 128
buf = torch.zeros(SLOTS * TILES * NST, dtype=torch.int64, device="cuda:0")
L = capi.load()
L.pies_exp_layer_stamps.argtypes = [ctypes.c_void_p]
assert L.pies_exp_layer_stamps(buf.data_ptr()) == 0
g.tick(2)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(SLOTS, TILES, NST)
assert L.pies_exp_layer_stamps(None) == 0
used = [s for s in range(SLOTS) if st[s, :, 1].any()]
print("launch slots with stamps: %d; config 2 after %d ticks; 20 x 20 x 250 beam, LAYERED" % (len(used), ticks))
for s in used[:6] + used[-2:]:
    tiles = np.nonzero(st[s, :, 1])[0]
    rows = st[s, tiles]
    counts = (rows != 0).sum(axis=1)
    n = int(np.bincount(counts).argmax())  # (a phase's first or last tile may hold one level only: fewer colours, fewer stamps)
    tiles, rows = tiles[counts == n], rows[counts == n]
    real0, t = rows[:, 0], rows[:, 1:n - 1].astype(np.float64)
    real1 = rows[:, n - 1]
    dur_real = (real1 - real0) * 10.0  # ns (100 MHz)
    cyc = t[:, -1] - t[:, 0]
    clock = np.median(cyc / np.maximum(dur_real, 1.0))  # GHz
    d = np.diff(t, axis=1) / clock  # ns
    med, p90, mx = np.median(d, axis=0), np.quantile(d, 0.9, axis=0), d.max(axis=0)
    span = (real1.max() - real0.min()) * 10.0
    print("slot %2d: %3d tiles, %2d stamps; in-kernel clock %.2f GHz; launch span (first tile start -> last tile end) %.2f us; "
          "tile lifetime median %.2f us max %.2f us; start skew (last tile start - first) %.2f us" % (
              s, len(tiles), n, clock, span / 1e3, np.median(dur_real) / 1e3, dur_real.max() / 1e3, (real0.max() - real0.min()) * 10.0 / 1e3))
    print("   intervals (ns) median over tiles: load %.0f | colours: %s | store %.0f" % (
        med[0], " ".join("%.0f" % v for v in med[1:-1]), med[-1]))
    print("   intervals (ns) 90th percentile : load %.0f | colours: %s | store %.0f" % (
        p90[0], " ".join("%.0f" % v for v in p90[1:-1]), p90[-1]))
g.close()
