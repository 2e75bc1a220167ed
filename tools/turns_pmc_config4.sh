#!/bin/bash
# Wave-state counters of the level launches of the reference order by turns (k_turn_round) on settled config 4: averages over the
# working launches.  On the GPU box: bash tools/turns_pmc_config4.sh > gpurun_out/r05_turns_pmc_config4.txt
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/c4tpmc; rm -rf $out; mkdir -p $out
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"; do
  d=$out/$(echo $pass | tr ' ' '_')
  (cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o t -- python3 $ROOT/tools/probe_turns.py 2 4 settled > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/t_counter_collection.csv")):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_turn_round" not in r["Kernel_Name"]: continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = list(disp.values())
    names = sorted(rows[0]) if rows else []
    key = "SQ_INSTS_VALU" if "SQ_INSTS_VALU" in names else "SQ_WAVE_CYCLES"
    floor = min(r.get(key, 0) for r in rows)
    work = [r for r in rows if r.get(key, 0) > 1.5 * floor + 1]
    print("%d k_turn_round launches, %d of them with work (more than 1.5 x the emptiest launch's %s)" % (len(rows), len(work), key))
    for n in names:
        v = sorted(r[n] for r in work)
        print("   %-22s median %12.0f   p10 %12.0f   p90 %12.0f   max %12.0f" % (n, v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10], v[-1]))
PY
rm -rf $out/*/t_kernel_trace.csv
