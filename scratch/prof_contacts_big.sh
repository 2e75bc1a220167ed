#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/contacts_big
export PIES_PROFILER_SAFE=1 PIES_NO_GRAPH=1 TICKS=6 PIES_TRI_FAST_ROWS=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/contacts_big/trace -- python scratch/pd_contacts_big.py > gpurun_out/contacts_big/out.txt 2> gpurun_out/contacts_big/err.txt; echo rc=$?
tail -1 gpurun_out/contacts_big/out.txt
find gpurun_out/contacts_big -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, os
f=max(glob.glob('gpurun_out/contacts_big/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-40s calls %6s avg %9.2f us  total %8.2f ms %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
