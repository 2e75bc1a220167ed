export PIES_PROFILER_SAFE=1 TMPDIR=/tmp
ROOT=$PWD; out=$ROOT/gpurun_out/c4tdur; rm -rf $out; mkdir -p $out
(cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $ROOT/tools/probe_turns.py 2 4 settled > $out/log.txt 2>&1) || { tail -5 $out/log.txt; exit 1; }
python3 - $out/t_kernel_trace.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[]; gaps=[]
prev_end=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if 'k_turn_round' in r['Kernel_Name']:
        d.append((e-s)/1e3)
        if prev_end is not None: gaps.append((s-prev_end)/1e3)
    prev_end=e
d.sort(); gaps.sort()
print("k_turn_round launches %d: duration us p10 %.1f median %.1f p90 %.1f max %.1f; gap before a launch us median %.1f p90 %.1f"%(len(d),d[len(d)//10],d[len(d)//2],d[9*len(d)//10],d[-1],gaps[len(gaps)//2],gaps[9*len(gaps)//10]))
fin=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'k_turn_finish' in r['Kernel_Name']]
print("k_turn_finish launches %d: total ms %.1f, longest ms %.1f"%(len(fin),sum(fin)/1e3,max(fin)/1e3 if fin else 0))
PY
rm -f $out/t_kernel_trace.csv
