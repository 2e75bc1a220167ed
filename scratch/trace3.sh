#!/bin/bash
# kernel-trace stats of some workloads (development aid): trace3.sh pd1m contacts ...
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PIES_PROFILER_SAFE=1
for W in "$@"; do
  OUT=gpurun_out/tr_$W; rm -rf $OUT
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/profile_target.py $W 6 > gpurun_out/tr_$W.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $W"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print("  %-46s calls %6s avg %8.1f us  %5.1f %%"%(r['Name'].replace('pies::','').split('(')[0][:46], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
