#!/bin/bash
# kernel trace of config 3 with a tuning at two values: prof_ab.sh NAME v0 v1 [workload]
set -e
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
W=${4:-config3}
ROOT=$PWD
for v in $2 $3; do
  out=$ROOT/gpurun_out/ab_$v
  rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $ROOT/tools/profile_target.py $W 12 $1=$v > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $1=$v"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total device us per tick: %.1f"%(tot/12/1e3))
for r in rows[:14]:
    print("  %-46s %5s %9.2f us %6s%%"%(r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY
done
