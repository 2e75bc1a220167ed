#!/bin/bash
# Wave-state counters of config 4's list-build launches in the settled state (the last tick's four builds).
# On the GPU box: bash tools/build_pmc_config4.sh > gpurun_out/r05_build_pmc_config4.txt
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/c4bpmc; rm -rf $out; mkdir -p $out
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  d=$out/$(echo $pass | tr ' ' '_')
  (cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o t -- python3 $ROOT/tools/profile_settled_config4.py > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/t_counter_collection.csv")):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_pair_build<384" not in r["Kernel_Name"]: continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)[-8:]
    for i in ids:
        if disp[i].get("SQ_WAVES", 1) == 0: continue
        print("dispatch %d: " % i + "  ".join("%s=%.0f" % kv for kv in sorted(disp[i].items())))
PY
rm -rf $out/*/t_kernel_trace.csv
