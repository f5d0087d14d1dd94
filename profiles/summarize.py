#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (written by profiles/collect.sh) into the tracked
summaries:  profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json and
profiles/pmc_traffic.json (the number bench.py quotes as roofline.traffic).

    python profiles/summarize.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
here = os.path.dirname(os.path.abspath(__file__))
raw = os.path.join(os.path.dirname(here), 'gpurun_out', tag)

KERNELS = ('k_pair_agg', 'k_pack', 'k_nosrc', 'k_cell_keys', 'k_cell_start')


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


stats = glob.glob(os.path.join(raw, 'stats', '**', '*_kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(here, '%s_kernel_stats.csv' % tag))

per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(raw, 'pmc_*', '**', '*_counter_collection.csv'), recursive=True):
    acc = defaultdict(float)          # (dispatch, kernel, counter) -> value
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k:
            acc[(r['Dispatch_Id'], k, r['Counter_Name'])] += float(r['Counter_Value'])
    for (_, k, c), v in acc.items():
        per[k][c].append(v)
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}

bench = json.loads(open(os.path.join(raw, 'bench.json')).read().strip().splitlines()[-1])
n = bench['config']['particles_per_gpu']
out = {
    'command': 'bash profiles/collect.sh %s  (rocprofv3 --pmc <one set> --kernel-trace '
               '--output-format csv -- python bench.py --steps 3 --warmup 1 '
               '--no-cpu-baseline; one pass per counter set)' % tag,
    'particles': n,
    'bench_line_same_box': bench,
    'per_launch_mean': mean,
}
cal = {}
KiB = 1024.0
if 'k_nosrc' in mean and 'FETCH_SIZE' in mean['k_nosrc']:
    cal['k_nosrc_fetch_B_per_particle (reads 8)'] = mean['k_nosrc']['FETCH_SIZE'] * KiB / n
    cal['k_nosrc_write_B_per_particle (writes 16)'] = mean['k_nosrc']['WRITE_SIZE'] * KiB / n
if 'k_cell_keys' in mean and 'FETCH_SIZE' in mean['k_cell_keys']:
    cal['k_cell_keys_fetch_B_per_particle (reads 24)'] = mean['k_cell_keys']['FETCH_SIZE'] * KiB / n
    cal['k_cell_keys_write_B_per_particle (writes 8)'] = mean['k_cell_keys']['WRITE_SIZE'] * KiB / n
out['calibration'] = cal
pa = mean.get('k_pair_agg', {})
if 'FETCH_SIZE' in pa and 'WRITE_SIZE' in pa:
    fetch = pa['FETCH_SIZE'] * KiB * 2.0      # gfx950: FETCH_SIZE reports 1/2 (guide + calibration above)
    write = pa['WRITE_SIZE'] * KiB
    out['k_pair_agg_traffic'] = {
        'fetch_bytes_corrected': fetch, 'write_bytes': write,
        'bytes_per_launch': fetch + write,
        'bytes_per_particle': (fetch + write) / n,
        'algorithmic_bytes_per_particle': 160.0,
    }
    if 'TCC_HIT_sum' in pa:
        out['k_pair_agg_traffic']['l2_hit_rate'] = pa['TCC_HIT_sum'] / (pa['TCC_HIT_sum'] + pa['TCC_MISS_sum'])
    json.dump({'config': {'n1': round(n ** (1 / 3.0)), 'variant': bench['config']['pair_variant'],
                          'spatially_ordered': bench['config']['spatially_ordered']},
               'bytes_per_launch': fetch + write,
               'source': 'profiles/%s_pmc_summary.json' % tag},
              open(os.path.join(here, 'pmc_traffic.json'), 'w'), indent=1)
json.dump(out, open(os.path.join(here, '%s_pmc_summary.json' % tag), 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'per_launch_mean'}, indent=1)[:3000])
