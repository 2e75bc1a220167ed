#!/bin/bash
# VALU instruction / busy counters of k_layer on BASELINE config 2 (100k beam, schedule LAYERED): is the launch bound by
# the issue rate of its projections' instruction streams?
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/valu_layer; mkdir -p $OUT
export PIES_PROFILER_SAFE=1
B="--no-cpu-baseline --no-exact --no-extras --no-scale --no-kernel-profile --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python bench.py $B > /dev/null 2> $OUT/a.err; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python bench.py $B > /dev/null 2> $OUT/b.err; echo rc=$?
python - <<'PY' | tee gpurun_out/valu_layer/summary.txt
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("gpurun_out/valu_layer/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k,v in acc.items():
            print(d,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()}, "dispatches", max(cnt[(k,c)] for c in v))
PY
find $OUT -name "*.csv" -size +4M -delete
