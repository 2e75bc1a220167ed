#!/bin/bash
# VALU instruction / busy counters of the PBD kernels at 1M particles (is k_tet arithmetic-bound there?)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/valu; mkdir -p $OUT
export PIES_PROFILER_SAFE=1
B="--no-cpu-baseline --no-exact --no-extras --no-scale --no-kernel-profile --dims 100 100 100 --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python bench.py $B > /dev/null 2> $OUT/a.err; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python bench.py $B > /dev/null 2> $OUT/b.err; echo rc=$?
python - <<'PY'
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("gpurun_out/valu/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            cnt[(k,r["Counter_Name"])]+=1
        for k,v in acc.items():
            print(d,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
