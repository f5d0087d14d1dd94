"""the figures of one bench.py line (python tools/debug/show_bench.py file.json)"""
import json, sys
d = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
for k, v in d['extra']['secondary'].items():
    print('%-70s %.3f' % (k[:70], v['ms_per_step']))
p = d['extra']['projected_strong_scaling_8']
print('slowest rank', p['slowest_rank_ms'], 'exchange', p['exchange_ms'], p['exchange_stand_in']['ms_per_step'], 't1', p['t_one_gpu_ms'],
      'free', p['projected_speedup_free_exchange'], {k: round(v['speedup'], 3) for k, v in p['projected_speedup_with_link'].items()})
print({r: round(v['ms_per_step'], 3) for r, v in p['ranks'].items()})
print({k: round(v['ms_per_step'], 3) for k, v in d['extra']['step_vs_n'].items()})
print(d['extra']['time_stepping']['ms_per_time_step'], d['extra']['parity_max_rel'], d['extra']['parity_ok'])
