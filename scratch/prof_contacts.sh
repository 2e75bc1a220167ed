#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/contacts
export PIES_PROFILER_SAFE=1 PIES_NO_GRAPH=1 PROF=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/contacts/trace -- python scratch/pd_contacts.py ${1:-10} ${2:-10} ${3:-60} > gpurun_out/contacts/out.txt 2> gpurun_out/contacts/err.txt; echo rc=$?
tail -2 gpurun_out/contacts/out.txt
find gpurun_out/contacts -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, os
f=max(glob.glob('gpurun_out/contacts/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:18]:
    print("%-40s calls %6s avg %9.2f us  total %8.2f ms %5s%%" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
