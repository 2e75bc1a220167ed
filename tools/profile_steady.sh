#!/bin/bash
# Kernel trace of the LAST ticks of a PD workload of tools/profile_target.py (the steady state instead of the onset the round
# profile shows): per-kernel averages per substep.  usage (GPU box): bash tools/profile_steady.sh pdcontacts 40 8 > gpurun_out/steady.txt
set -e
W=${1:-pdcontacts}; N=${2:-40}; LAST=${3:-8}; EXTRA=${4:-}
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/steady_$W; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $ROOT/tools/profile_target.py $W $N $EXTRA > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
python3 - $out/t_kernel_trace.csv $LAST <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1]))); last=int(sys.argv[2])
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_pd_predict' in r['Kernel_Name'] or 'k_predict' in r['Kernel_Name']]
seg=rows[idx[-last-1]:idx[-1]]
dur=collections.defaultdict(float); cnt=collections.Counter()
for r in seg:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('pies::','')[:40]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    dur[n]+=d; cnt[n]+=1
print("last %d substeps of %d: device time per substep %.1f us, wall %.1f us"%(last,len(idx),sum(dur.values())/last,(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3/last))
for n in sorted(dur,key=lambda n:-dur[n])[:18]:
    print("  %-42s x%6.1f  avg %8.2f  per substep %8.1f"%(n,cnt[n]/last,dur[n]/cnt[n],dur[n]/last))
PY
rm -f $out/t_kernel_trace.csv
