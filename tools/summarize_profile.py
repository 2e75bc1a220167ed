#!/usr/bin/env python3
"""Copies the judged summaries of one profiling round from gpurun_out/<round>/ (scratch) into profiles/
(tracked): the rocprofv3 --kernel-trace --stats table, the bench lines, and the per-kernel HBM traffic
derived from the two PMC passes as MI355X_MICROARCH.md prescribes for gfx950:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (FETCH_SIZE counts 64 B per 128-B request, in KiB)

usage: python tools/summarize_profile.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name.split("::")[-1].split("<")[0]


def main(rnd):
    src = os.path.join(ROOT, "gpurun_out", rnd)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)  # gpurun merges runs into the same directory
    stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % rnd))
    for f in ("bench_plain.json", "bench_traced.json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (rnd, f)))
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind in ("pmc_fetch", "pmc_write"):
        for f in [newest(os.path.join(src, kind, "*", "*counter_collection.csv"))]:
            for r in csv.DictReader(open(f)):
                counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    for k, c in sorted(counters.items()):
        fetch = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
        write = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
        traffic[k] = {"FETCH_SIZE_KiB_avg": fetch, "WRITE_SIZE_KiB_avg": write, "dispatches": len(c["FETCH_SIZE"]),
                      "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0}
    with open(os.path.join(dst, "%s_pmc_traffic.json" % rnd), "w") as f:
        json.dump(traffic, f, indent=1)
    shutil.copy(os.path.join(dst, "%s_pmc_traffic.json" % rnd), os.path.join(dst, "pmc_traffic.json"))  # read by bench.py
    rows = list(csv.DictReader(open(stats)))
    print("kernel                 calls   avg_us   pct     hbm_bytes/launch")
    for r in rows:
        k = short(r["Name"])
        print("%-20s %7s %8.2f %6s %14.0f" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"],
                                                traffic.get(k, {}).get("hbm_bytes_per_launch", float("nan"))))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
