import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(k):
    for n,v in k.items(): print("   %-18s %4d x %9.2f us  units %9.0f  %7.0f GB/s  %.3f"%(n,v['launches_per_substep'],v['avg_us'],v['units_per_launch'],v['algorithmic_GBs'],v['hbm_frac']))
print('config2', d['value']); show(d['kernels'])
if 'other_configs' in d:
    print('pd3', d['other_configs']['pd_config3']['value']); show(d['other_configs']['pd_config3']['kernels'])
    print('c4', d['other_configs']['collisions_config4']['value'])
if 'scale_1m' in d:
    s=d['scale_1m']
    print('pbd1m', s['pbd_1m']['substeps_per_sec'], s['pbd_1m']['launches_per_substep']); show(s['pbd_1m']['kernels'])
    print('pd1m', s['pd_1m']['substeps_per_sec'], s['pd_1m']['pcg_stats']); show(s['pd_1m']['kernels'])
if 'exact_order' in d: print('exact', d['exact_order']['value'])
if 'cpu_baseline' in d: print('cpu', d['cpu_baseline']['value'])
