#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tri
export PIES_PROFILER_SAFE=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tri/trace -- python scratch/tri_perf.py 20 > gpurun_out/tri/out.txt 2> gpurun_out/tri/err.txt; echo rc=$?
cat gpurun_out/tri/out.txt
find gpurun_out/tri -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, os
f=max(glob.glob('gpurun_out/tri/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-40s calls %6s avg %9.2f us  total %9.2f ms %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
