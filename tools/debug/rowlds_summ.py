"""per-launch means of the pair kernels of the S-rings with and without option row_lds (gpurun_out/r06_rowlds_{on,off},
written by profiles/collect.sh): duration from the kernel trace, every PMC pass"""
import csv, glob, os, sys
from collections import defaultdict
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'gpurun_out')
def short(n):
    if 'k_pair_rowlds' in n: return 'rates (row tiles in LDS)'
    if 'k_pair_wave' in n and 'FamElastic' in n: return 'rates (wave kernel)'
    if 'k_pair_wave' in n and 'FamVGrad' in n: return 'velocity gradient'
    return None
for tag in ('r06_rowlds_off', 'r06_rowlds_on'):
    d = os.path.join(root, tag, 'custom')
    print('==', tag)
    dur = defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, 'stats', 'runc_kernel_trace.csv'))):
        s = short(r['Kernel_Name'])
        if s: dur[s].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    for s, v in dur.items():
        v = v[6:] if len(v) > 6 else v
        print('  %-28s %4d launches  mean %.1f us' % (s, len(v), sum(v) / len(v)))
    cnt = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, 'pmc_*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            s = short(r['Kernel_Name'])
            if s: cnt[s][r['Counter_Name']].append(float(r['Counter_Value']))
        for r in csv.DictReader(open(f)):
            extra = {k: r[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'SGPR_Count') if k in r}
            s = short(r['Kernel_Name'])
            if s and s not in cnt['_res']: cnt['_res'][s] = extra
    for s in dur:
        print('  ', s, dict(cnt['_res'].get(s, {})))
        for k in sorted(cnt[s]):
            v = cnt[s][k]
            print('      %-40s %.4g' % (k, sum(v) / len(v)))
