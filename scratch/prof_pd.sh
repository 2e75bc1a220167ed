#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pd
export PIES_PROFILER_SAFE=1
W=${1:-10000}; CG=${2:-48}
timeout 200 python scratch/pd_bench.py 20 20 250 10 $CG $W
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd/trace -- python scratch/pd_bench.py 20 20 250 4 $CG $W > gpurun_out/pd/out.txt 2> gpurun_out/pd/err.txt; echo rc=$?
find gpurun_out/pd -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, os
f=max(glob.glob('gpurun_out/pd/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:30]:
    print("%-40s calls %6s avg %9.2f us  total %8.2f ms %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
