#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pd
export PIES_PROFILER_SAFE=1
timeout 200 python scratch/pd_bench.py 20 20 250 20 32
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd/trace -- python scratch/pd_bench.py 20 20 250 6 32 > gpurun_out/pd/out.txt 2> gpurun_out/pd/err.txt; echo rc=$?
find gpurun_out/pd -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/pd/trace/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print("%-40s calls %6s avg %9.2f us  %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
