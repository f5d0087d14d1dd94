"""print the BASELINE.md section-4 tables from a `python bench.py` JSON line (profiles/rNN_bench_default.json)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d['extra']
def row(name, n, dtype, v):
    k = v['kernel_ms_per_step']
    fam = v.get('pair_ms_per_family') or {}
    fams = ' + '.join('%.2f' % x for x in fam.values()) if len(fam) > 1 else '%.3f' % k['pair']
    par = '–' if v.get('parity_max_rel') is None else '%.1e / %.1e / %d' % (
        v['parity_max_rel'], v.get('parity_elementwise_max_rel', float('nan')), v.get('parity_neighbour_count_mismatches', 0))
    fr = v.get('roofline', {}).get('frac')
    print('| %s | %s | %s | %.3g | %.2f | %.3f / %.3f / %.3f / %s | %s | %s |' % (
        name, format(n, ',').replace(',', ' '), dtype, v.get('particle_updates_per_s', 0), v['ms_per_step'],
        k['nnps'], k['pack'], k['eos'], fams, '%.1f %%' % (100 * fr) if fr else '', par))
print('| Config | particles | dtype | particle-updates/s | ms/step | nnps / pack / eos / pair | pair passes, % of 8 TB/s by algorithmic bytes | vs oracle at this size: norm-wise / element-wise / count mismatches |')
print('|---|---|---|---|---|---|---|---|')
head = {'kernel_ms_per_step': d['kernel_ms_per_step'], 'ms_per_step': d['ms_per_step'], 'particle_updates_per_s': d['value'],
        'roofline': d['roofline'], 'parity_max_rel': e.get('parity_max_rel'), 'parity_elementwise_max_rel': e.get('parity_elementwise_max_rel'),
        'parity_neighbour_count_mismatches': e.get('parity_neighbour_count_mismatches')}
row('**S-cube WCSPH (headline)**', d['config']['particles_per_gpu'], d['dtype'], head)
for k, v in e['secondary'].items():
    row(k, v.get('particles', 0), v.get('dtype', 'f64'), v)
print()
print('| particles | ms/step | updates/s | nnps | pack | eos | pair |')
print('|---|---|---|---|---|---|---|')
for k, v in e['step_vs_n'].items():
    kk = v['kernel_ms_per_step']
    print('| %s (%s) | %.3f | %.3g | %.3f | %.3f | %.3f | %.3f |' % (format(v['particles'], ',').replace(',', ' '), k, v['ms_per_step'],
          v['particle_updates_per_s'], kk['nnps'], kk['pack'], kk['eos'], kk['pair']))
p = e.get('projected_strong_scaling_8')
if p:
    print()
    print('| rank | real particles (fluid / boundary / obstacle) | ghosts | ms/step | nnps | pack | eos | pair | bytes received per face |')
    print('|---|---|---|---|---|---|---|---|---|')
    for r, v in p['ranks'].items():
        if 'ms_per_step' not in v:
            continue
        kk = v['kernel_ms_per_step']
        ra = v['real_per_array']
        print('| %s | %s / %s / %s | %d k | %.3f | %.3f | %.3f | %.3f | %.3f | %s |' % (
            r, format(ra.get('fluid', 0), ',').replace(',', ' '), format(ra.get('boundary', 0), ',').replace(',', ' '),
            format(ra.get('obstacle', 0), ',').replace(',', ' '), v['ghost_particles'] // 1000, v['ms_per_step'], kk['nnps'], kk['pack'],
            kk['eos'], kk['pair'], ' / '.join('%.1f MB' % (b / 1e6) for b in v['bytes_per_face'])))
    print()
    print(json.dumps({k: v for k, v in p.items() if k != 'ranks'}, indent=1))
print(e.get('time_stepping'))
print(d.get('cpu_baseline'), e.get('cpu_baseline_100'))
