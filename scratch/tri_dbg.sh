#!/bin/bash
# kernels of the triangle pipeline under tuning settings (development aid): tri_dbg.sh "NAME=V" "NAME=V" ...
export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD
for m in "$@"; do
  out=$ROOT/gpurun_out/tri_dbg; rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $ROOT/scratch/tri_dbg.py $m > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
  python3 - $out/t_kernel_trace.csv "$m" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_pd_predict' in r['Kernel_Name']]
seg=rows[idx[-6]:idx[-1]]
dur=collections.defaultdict(float); cnt=collections.Counter()
for r in seg:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('pies::','')[:34]
    dur[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; cnt[n]+=1
print(sys.argv[2], "total %.1f"%(sum(dur.values())/5), " ".join("%s %.1f"%(n[2:],dur[n]/cnt[n]) for n in sorted(dur) if 'k_tri' in n))
PY
done
