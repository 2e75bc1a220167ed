#!/bin/bash
# Counters of the co-residency micro-benchmark of the PBD tetrahedral projection (tools/svd_overlap.hip, built to
# tools/_bin/svd_overlap): one workgroup per compute unit of 256 / 512 / 768 / 1 024 threads = 1 / 2 / 3 / 4 wavefronts per SIMD
# running the SAME dependent chain of 200 projections per lane.  VERDICT r4 item 5 asked for SQ_INSTS_VALU, SQ_BUSY_CYCLES,
# SQ_WAIT_INST_ANY of both variants.  On the GPU box: bash tools/svd_overlap_pmc.sh > gpurun_out/r05_svd_overlap_pmc.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/svdpmc; rm -rf $out; mkdir -p $out
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  d=$out/$(echo $pass | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o t -- $R/tools/_bin/svd_overlap > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  grep "wavefront(s) per SIMD" $d.log | head -4
done
cd $R && python3 - <<'PY'
import csv, glob, collections
tab = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/svdpmc/*/t_counter_collection.csv")):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_chain" not in r["Kernel_Name"]: continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"threads": int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 0})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in disp.values():  # (six launches per size: one warm-up + five timed: the last one of a size is kept)
        tab.setdefault(d["threads"], {}).update({k: v for k, v in d.items() if k != "threads"})
names = sorted({k for v in tab.values() for k in v})
print("%-10s %s" % ("threads", " ".join("%20s" % n for n in names)))
for t, v in tab.items():
    print("%-10d %s" % (t, " ".join("%20.0f" % v.get(n, float("nan")) for n in names)))
PY
