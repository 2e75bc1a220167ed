#!/bin/bash
# kernel stats of config 4 (500k loose particles, node-node collisions)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/coll
export PIES_PROFILER_SAFE=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/coll/trace -- python scratch/coll_prof.py > gpurun_out/coll/out.txt 2> gpurun_out/coll/err.txt; echo rc=$?
tail -12 gpurun_out/coll/out.txt
find gpurun_out/coll -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, os
f=max(glob.glob('gpurun_out/coll/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-40s calls %6s avg %9.2f us  total %8.2f ms %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
