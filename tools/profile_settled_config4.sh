#!/bin/bash
# Kernel trace of config 4 once the burst is over (ticks 9-11 of 12, 64 captured level launches): per-kernel averages.
# On the GPU box: bash tools/profile_settled_config4.sh > gpurun_out/r03_settled_config4.txt (copied to profiles/ afterwards).
set -e
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/c4s; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $ROOT/tools/profile_settled_config4.py > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
tail -1 $out/log.txt
python3 - $out/t_kernel_trace.csv <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_predict' in r['Kernel_Name']]
seg=rows[idx[-4]:idx[-1]]   # three whole ticks of the settled state
T=3
dur=collections.defaultdict(float); cnt=collections.Counter(); work=collections.defaultdict(float); wcnt=collections.Counter()
for r in seg:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('pies::','')[:34]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    dur[n]+=d; cnt[n]+=1
    if d>6: work[n]+=d; wcnt[n]+=1
print("per tick (us): total %.1f   wall %.1f"%(sum(dur.values())/T,(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3/T))
for n in sorted(dur,key=lambda n:-dur[n])[:22]:
    print("  %-36s x%6.1f  avg %8.2f  per tick %8.1f   (>6us: x%.1f avg %.1f)"%(n,cnt[n]/T,dur[n]/cnt[n],dur[n]/T,wcnt[n]/T,work[n]/max(1,wcnt[n])))
# the level launches of the last pass, in order
last=[r for r in seg if 'k_pair_round' in r['Kernel_Name'] or 'k_pair_save' in r['Kernel_Name']]
cut=max(i for i,r in enumerate(last) if 'k_pair_save' in r['Kernel_Name'])
lv=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in last[cut+1:]]
print("level launches of the last pass (us):"," ".join("%.0f"%d for d in lv))
PY

rm -f $out/t_kernel_trace.csv
