#!/usr/bin/env python3
"""Copies the judged summaries of one profiling round from gpurun_out/<round>/ (scratch, produced on the GPU box by
tools/profile_round.sh) into profiles/ (tracked):

  <round>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of whole substeps of the workload (in situ)
  <round>_pmc_traffic.json              per workload and kernel: HBM bytes per launch from the two --pmc passes, corrected as
                                        MI355X_MICROARCH.md prescribes for gfx950:
                                            hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (FETCH_SIZE counts 64 B per 128-B request, KiB)
  <round>_pmc_valu_config2.json         VALU / wave counters per launch of the dominant kernel
  <round>_summary.txt                   one table per workload: calls, average duration, share of device time, HBM bytes per launch

usage: python tools/summarize_profile.py r02 [source-directory-name]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("config2", "config3", "config4", "contacts", "pdcontacts", "pbd1m", "pd1m", "pd1m_work", "pd1m_streamed", "pd_unstructured",
             "pd_unstructured_1m")


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name.split("::")[-1].split("<")[0]


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def counters_of(path):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if path:
        for r in csv.DictReader(open(path)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main(rnd, srcname=None):
    src = os.path.join(ROOT, "gpurun_out", srcname or rnd)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    traffic = {}
    lines = []
    for w in WORKLOADS:
        stats = newest(os.path.join(src, "trace_%s" % w, "*", "*kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, w)))
        fetch = counters_of(newest(os.path.join(src, "fetch_%s" % w, "*", "*counter_collection.csv")))
        write = counters_of(newest(os.path.join(src, "write_%s" % w, "*", "*counter_collection.csv")))
        traffic[w] = {}
        for k in sorted(set(fetch) | set(write)):
            f = fetch[k]["FETCH_SIZE"]
            wr = write[k]["WRITE_SIZE"]
            fa = sum(f) / max(1, len(f))
            wa = sum(wr) / max(1, len(wr))
            traffic[w][k] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "dispatches": max(len(f), len(wr)),
                             "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
        rows = list(csv.DictReader(open(stats)))
        total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
        lines.append("== %s  (rocprofv3 --kernel-trace --stats -- python3 tools/profile_target.py %s N; PIES_PROFILER_SAFE=1)" % (w, w))
        lines.append("%-28s %8s %12s %8s %18s" % ("kernel", "calls", "avg_us", "share", "hbm_bytes/launch"))
        for r in rows:
            k = short(r["Name"])
            t = traffic[w].get(k, {}).get("hbm_bytes_per_launch")
            lines.append("%-28s %8s %12.2f %7.2f%% %18s" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / total,
                                                          "%.0f" % t if t is not None else "-"))
        if w == "config4":
            lines.append("(the first six ticks = the burst.  Under the profiler the library does not re-capture its graph, so the captured level")
            lines.append(" launches stay at their initial count and the burst's deeper passes are finished behind the grid barrier of k_pair_repeat -")
            lines.append(" an artefact of PIES_PROFILER_SAFE (in a run the host re-captures with the deepest pass + 8 launches); the settled state, level by")
            lines.append(" level: %s_settled_config4.txt, %s_levels_pmc_config4.txt, %s_build_pmc_config4.txt)" % (rnd, rnd, rnd))
        lines.append("")
    with open(os.path.join(dst, "%s_pmc_traffic.json" % rnd), "w") as f:
        json.dump(traffic, f, indent=1)
    valu = counters_of(newest(os.path.join(src, "valu_config2", "*", "*counter_collection.csv")))
    if valu:
        out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in valu.items()}
        with open(os.path.join(dst, "%s_pmc_valu_config2.json" % rnd), "w") as f:
            json.dump(out, f, indent=1)
        lines.append("== config2 VALU counters per launch (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY)")
        for k, cs in sorted(out.items()):
            lines.append("%-28s %s" % (k, "  ".join("%s=%.0f" % kv for kv in sorted(cs.items()))))
    valu4 = counters_of(newest(os.path.join(src, "valu_config4", "*", "*counter_collection.csv")))
    if valu4:
        out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in valu4.items()}
        with open(os.path.join(dst, "%s_pmc_valu_config4.json" % rnd), "w") as f:
            json.dump(out, f, indent=1)
        lines.append("")
        lines.append("== config4 VALU counters per launch, averaged over the launches of a kernel (same counters)")
        for k, cs in sorted(out.items()):
            lines.append("%-28s %s" % (k, "  ".join("%s=%.0f" % kv for kv in sorted(cs.items()))))
    # VALU counters of the PD workloads (k_pd_local_tiles): <round>_pmc_valu_pd.json = {workload: {kernel: {counter: average per launch}}}
    pd = {}
    for w in ("config3", "pd1m", "pd1m_work", "pd_unstructured"):
        cs = counters_of(newest(os.path.join(src, "valu_%s" % w, "*", "*counter_collection.csv")))
        if cs:
            pd[w] = {k: {c: sum(v) / len(v) for c, v in kc.items()} for k, kc in cs.items()}
            lines.append("")
            lines.append("== %s VALU counters per launch (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY)" % w)
            for k, kc in sorted(pd[w].items()):
                lines.append("%-28s %s" % (k, "  ".join("%s=%.0f" % kv for kv in sorted(kc.items()))))
    if pd:
        with open(os.path.join(dst, "%s_pmc_valu_pd.json" % rnd), "w") as f:
            json.dump(pd, f, indent=1)
    text = "\n".join(lines) + "\n"
    with open(os.path.join(dst, "%s_summary.txt" % rnd), "w") as f:
        f.write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03", sys.argv[2] if len(sys.argv) > 2 else None)
