"""rewrite section 4 of BASELINE.md from profiles/r06_bench_default.json (tools/baseline_table.py makes the tables)"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, 'profiles', 'r06_bench_default.json')
tables = subprocess.check_output([sys.executable, os.path.join(REPO, 'tools', 'baseline_table.py'), src], universal_newlines=True).split('\n\n')
t_main, t_n, t_ranks = tables[0], tables[1], tables[2]
d = json.loads(open(src).read().strip().splitlines()[-1])
pj = d['extra']['projected_strong_scaling_8']
wl = pj['projected_speedup_with_link']
ts = d['extra']['time_stepping']
sec = {k.split(' ')[0] + ' ' + k.split(' ')[1] if False else k: v for k, v in d['extra']['secondary'].items()}
def ms(prefix):
    for k, v in sec.items():
        if k.startswith(prefix):
            return v['ms_per_step']
    return float('nan')
T = open(os.path.join(REPO, 'tools', 'baseline_section.tmpl')).read()
rows = ''
for k in ('45 GB/s', '60 GB/s', '75 GB/s'):
    rows += '| %s | %.3f ms | %.3f ms | **%.2f×** | %.2f× |\n' % (k, wl[k]['transfer_ms'], wl[k]['step_ms'], wl[k]['speedup'], wl[k]['speedup_round5_messages'])
rows += '| (free exchange) | – | %.3f ms | %.2f× | |\n' % (pj['slowest_rank_ms'], pj['projected_speedup_free_exchange'])
rep = {
    '{T_MAIN}': t_main, '{T_N}': t_n, '{T_RANKS}': t_ranks, '{T_LINK}': rows,
    '{TS_MS}': '%.2f' % ts['ms_per_time_step'], '{TS_PS}': '%.0f' % ts['time_steps_per_s'],
    '{CPU}': '%.2e' % d['cpu_baseline']['value'], '{RATIO}': '%.0f' % (d['value'] / d['cpu_baseline']['value']),
    '{EX_MS}': '%.3f' % pj['exchange_ms'], '{FACE_MB}': '%.1f' % (pj['largest_face_message_bytes'] / 1e6),
    '{FACE5_MB}': '%.1f' % (pj['largest_face_message_bytes_round5_protocol'] / 1e6),
    '{T1}': '%.2f' % pj['t_one_gpu_ms'], '{T6}': '%.2f' % (pj['t_one_gpu_ms'] / 6.0), '{SLOW}': '%.2f' % pj['slowest_rank_ms'],
    '{HEAD}': '%.2f' % d['ms_per_step'], '{PAIR}': '%.3f' % d['roofline']['avg_kernel_ms'], '{FRAC}': '%.1f' % (100 * d['roofline']['frac']),
    '{N1M}': '%.2f' % ms('100^3'), '{C2}': '%.2f' % ms('C2 '), '{D4}': '%.2f' % ms('dam break dx 0.0055 (4 M'), '{D16}': '%.2f' % ms('C4 '),
    '{TG}': '%.2f' % ms('C3 '), '{R32}': '%.2f' % ms('C5 S-rings3d 2 M fp32'), '{R64}': '%.2f' % ms('C5 S-rings3d 2 M fp64'),
}
for k, v in rep.items():
    T = T.replace(k, v)
p = os.path.join(REPO, 'BASELINE.md')
s = open(p).read()
a = s.index('## 4. Results (round')
open(p, 'w').write(s[:a] + T)
print('BASELINE.md section 4 rewritten from', src)
