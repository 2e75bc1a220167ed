#!/bin/bash
# Kernel trace of config 3 in the state bench.py times (the beam after 34 frames, swinging): per-kernel averages of the last five ticks.
# On the GPU box: bash tools/profile_moving_config3.sh > gpurun_out/r03_moving_config3.txt (copied to profiles/ afterwards).
set -e
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/moving_config3; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $ROOT/tools/profile_moving_config3.py > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
python3 - $out/t_kernel_trace.csv <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_pd_predict' in r['Kernel_Name']]
seg=rows[idx[-6]:idx[-1]]   # five whole ticks of the moving state
dur=collections.defaultdict(float); cnt=collections.Counter()
for r in seg:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('pies::','')[:34]
    dur[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; cnt[n]+=1
print("per tick (us): total %.1f"%(sum(dur.values())/5))
for n in sorted(dur,key=lambda n:-dur[n]):
    print("  %-36s x%5.1f  avg %7.2f  per tick %7.1f"%(n,cnt[n]/5,dur[n]/cnt[n],dur[n]/5))
PY
rm -f $out/t_kernel_trace.csv
