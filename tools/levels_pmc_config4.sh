#!/bin/bash
# Wave-state counters of config 4's level launches in the settled state, level by level (the last pass of tick 12):
# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (MI355X_MICROARCH.md).
# On the GPU box: bash tools/levels_pmc_config4.sh > gpurun_out/r05_levels_pmc_config4.txt
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/c4pmc; rm -rf $out; mkdir -p $out
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  d=$out/$(echo $pass | tr ' ' '_')
  (cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o t -- python3 $ROOT/tools/profile_settled_config4.py > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
table = collections.OrderedDict()
names = []
for f in sorted(glob.glob(out + "/*/t_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    saves = [i for i in ids if "k_pair_save" in disp[i]["k"]]
    levels = [i for i in ids if i > saves[-1] and "k_pair_round" in disp[i]["k"]]
    for n, i in enumerate(levels):
        row = table.setdefault(n + 1, {})
        for c, v in disp[i].items():
            if c != "k":
                row[c] = v
                if c not in names: names.append(c)
print("level " + " ".join("%18s" % c for c in names))
for n, row in table.items():
    print("%5d " % n + " ".join("%18.0f" % row.get(c, float("nan")) for c in names))
PY
rm -rf $out/*/t_kernel_trace.csv
