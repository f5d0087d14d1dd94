#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (written by profiles/collect.sh) into the tracked
summaries:

    profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (top kernels)
    profiles/<tag>_pmc_summary.json              per-launch means of every PMC pass (cube workload),
                                                 FETCH/WRITE calibration, derived HBM traffic
    profiles/<tag>_bench_lines.json              the bench line of every profiled workload (same box)
    profiles/pmc_traffic.json                    the number bench.py quotes as roofline.traffic

    python profiles/summarize.py r02
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
here = os.path.dirname(os.path.abspath(__file__))
raw = os.path.join(os.path.dirname(here), 'gpurun_out', tag)

KERNELS = ('k_pair_wave', 'k_pack', 'k_nosrc', 'k_bin_keys', 'k_bucket_scatter', 'k_bucket_sort')


def short(name):
    for k in KERNELS:
        if k in name:
            if k == 'k_pair_wave':
                fam = name.split('<')[1].split(',')[0].replace('_T', '')
                return 'k_pair_wave<%s>' % fam
            return k
    return None


lines = {}
for wdir in sorted(glob.glob(os.path.join(raw, '*'))):
    w = os.path.basename(wdir)
    if not os.path.isdir(wdir):
        continue
    stats = glob.glob(os.path.join(wdir, 'stats', '**', '*_kernel_stats.csv'), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(here, '%s_%s_kernel_stats.csv' % (tag, w)), 'w') as f:
            wr = csv.writer(f)
            wr.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
            for r in rows[:14]:
                wr.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                             r['MinNs'], r['MaxNs'], r['Percentage']])
    try:
        lines[w] = json.loads(open(os.path.join(wdir, 'bench.json')).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(lines, open(os.path.join(here, '%s_bench_lines.json' % tag), 'w'), indent=1)

KiB = 1024.0


def pmc_means(wdir):
    """per-launch mean of every counter of every hot kernel of one workload"""
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(wdir, 'pmc_*', '**', '*_counter_collection.csv'), recursive=True):
        acc = defaultdict(float)          # (dispatch, kernel, counter) -> value
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                acc[(r['Dispatch_Id'], k, r['Counter_Name'])] += float(r['Counter_Value'])
        for (_, k, c), v in acc.items():
            per[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}


def derive(m, n, algo=None):
    """what the counters of one pair kernel say, per launch (n particles)"""
    tr = {}
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        # gfx950: FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM
        # section; reproduced by the calibration kernels), WRITE_SIZE is exact
        fetch = m['FETCH_SIZE'] * KiB * 2.0
        write = m['WRITE_SIZE'] * KiB
        tr.update({'fetch_bytes_corrected': fetch, 'write_bytes': write, 'bytes_per_launch': fetch + write,
                   'bytes_per_particle': (fetch + write) / n if n else None})
        if algo:
            tr['algorithmic_bytes_per_particle'] = algo
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
        tr['l2_hit_rate'] = m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
    if 'GRBM_GUI_ACTIVE' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0          # summed over the 8 XCDs
        tr['kernel_cycles_per_xcd'] = cyc
        if 'SQ_ACTIVE_INST_VALU' in m:
            tr['valu_busy'] = m['SQ_ACTIVE_INST_VALU'] * 4.0 / (cyc * 1024.0)
        if 'TA_BUSY_avr' in m:
            tr['ta_busy'] = m['TA_BUSY_avr'] / cyc
        if 'TCP_PENDING_STALL_CYCLES_sum' in m:
            tr['tcp_pending_stall_frac'] = m['TCP_PENDING_STALL_CYCLES_sum'] / 256.0 / cyc
        if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m:
            tr['tcp_accesses_per_cu_cycle'] = m['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256.0 / cyc
        if 'TCP_READ_TAGCONFLICT_STALL_CYCLES_sum' in m:
            tr['tcp_tagconflict_stall_frac'] = m['TCP_READ_TAGCONFLICT_STALL_CYCLES_sum'] / 256.0 / cyc
    if 'TCP_TCC_READ_REQ_sum' in m:
        # 128-B line requests of the CUs' L1s to the L2 (calibration: k_nosrc reads 8 B per
        # particle coalesced and counts 8/128 requests per particle, k_cell_keys 24/128)
        tr['l1_fill_bytes_per_launch'] = m['TCP_TCC_READ_REQ_sum'] * 128.0
        tr['l1_fill_bytes_per_particle'] = m['TCP_TCC_READ_REQ_sum'] * 128.0 / n if n else None
        if 'GRBM_GUI_ACTIVE' in m:
            tr['l1_fill_bytes_per_cu_cycle'] = m['TCP_TCC_READ_REQ_sum'] * 128.0 / 256.0 / (m['GRBM_GUI_ACTIVE'] / 8.0)
    if 'SQ_INSTS_VALU' in m and 'SQ_WAVES' in m:
        tr['valu_insts_per_wave'] = m['SQ_INSTS_VALU'] / m['SQ_WAVES']
        tr['vmem_rd_per_wave'] = m['SQ_INSTS_VMEM_RD'] / m['SQ_WAVES']
    return tr


summary = {'command': 'bash profiles/collect.sh %s  (rocprofv3 --pmc <one set> --kernel-trace '
                      '--output-format csv -- python bench.py --no-cpu-baseline --no-check --no-extras '
                      '--steps 3 --warmup 1 [workload flags]; one pass per counter set)' % tag,
           'workloads': {}}
for w, bench in lines.items():
    wdir = os.path.join(raw, w)
    mean = pmc_means(wdir)
    if not mean:
        continue
    n = bench.get('config', {}).get('particles_per_gpu', 0)
    entry = {'particles': n, 'bench_line_same_box': bench, 'per_launch_mean': mean, 'pair_kernels': {}}
    cal = {}
    if n and 'k_nosrc' in mean and 'FETCH_SIZE' in mean['k_nosrc'] and w.startswith('cube'):
        cal['k_nosrc_fetch_B_per_particle (reads 8)'] = mean['k_nosrc']['FETCH_SIZE'] * KiB / n
        cal['k_nosrc_write_B_per_particle (writes 16)'] = mean['k_nosrc']['WRITE_SIZE'] * KiB / n
    if n and 'k_bin_keys' in mean and 'FETCH_SIZE' in mean['k_bin_keys']:
        cal['k_bin_keys_fetch_B_per_particle (reads 24)'] = mean['k_bin_keys']['FETCH_SIZE'] * KiB / n
        cal['k_bin_keys_write_B_per_particle (writes 4)'] = mean['k_bin_keys']['WRITE_SIZE'] * KiB / n
    for kk in ('k_nosrc', 'k_bin_keys'):
        if n and kk in mean and 'TCP_TCC_READ_REQ_sum' in mean[kk]:
            cal['%s_l1_line_requests_per_particle (x128 B)' % kk] = mean[kk]['TCP_TCC_READ_REQ_sum'] / n
    entry['calibration'] = cal
    for k in sorted(mean):
        if k.startswith('k_pair_wave'):
            algo = bench.get('roofline', {}).get('algorithmic_bytes_per_particle') if len(
                [q for q in mean if q.startswith('k_pair_wave') and 'FamNbr' not in q]) == 1 and 'FamNbr' not in k else None
            entry['pair_kernels'][k] = derive(mean[k], n, algo)
    summary['workloads'][w] = entry
    if w == 'cube' and entry['pair_kernels']:
        # the equation kernel, not the neighbour-count pass bench.py runs once outside the timed loop
        tr = [v for k, v in sorted(entry['pair_kernels'].items()) if 'FamNbr' not in k][0]
        if 'bytes_per_launch' in tr:
            json.dump({'config': {'n1': 159, 'variant': bench['config']['pair_variant'],
                                  'spatially_ordered': bench['config']['spatially_ordered'],
                                  'workload': 'cube', 'dtype': bench.get('dtype', 'f64')},
                       'bytes_per_launch': tr['bytes_per_launch'],
                       'l1_fill_bytes_per_launch': tr.get('l1_fill_bytes_per_launch'),
                       # the pair kernel's live HIP-event mean of the un-profiled bench run on the box the counters
                       # were taken on (same call): bench.py quotes it next to its own kernel time
                       'profiled_box_kernel_ms': bench.get('roofline', {}).get('avg_kernel_ms'),
                       'source': 'profiles/%s_pmc_summary.json: FETCH_SIZE x 2 + WRITE_SIZE of separate '
                                 'rocprofv3 --pmc passes of this command on another box, not measured in '
                                 'this run' % tag},
                      open(os.path.join(here, 'pmc_traffic.json'), 'w'), indent=1)
json.dump(summary, open(os.path.join(here, '%s_pmc_summary.json' % tag), 'w'), indent=1)
for w, e in summary['workloads'].items():
    print(w, json.dumps(e['pair_kernels'], indent=1))
    print(json.dumps(e['calibration'], indent=1))
