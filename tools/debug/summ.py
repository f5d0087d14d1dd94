import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],4), 'value', '%.4g' % d['value'])
print('kernel_ms', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','avg_kernel_ms','traffic','traffic_profiled_kernel_ms')})
print('cpu', d.get('cpu_baseline'))
e=d.get('extra',{})
print({k: e[k] for k in e if k.startswith('parity')})
for k,v in e.get('secondary',{}).items():
    print(k[:58], round(v['ms_per_step'],3), {a:round(b,3) for a,b in v['kernel_ms_per_step'].items()}, {a: round(b,3) for a,b in v.get('pair_ms_per_family',{}).items()}, v.get('parity_max_rel'), v.get('roofline',{}).get('frac'))
for k,v in e.get('step_vs_n',{}).items():
    print(k, round(v['ms_per_step'],3), {a:round(b,3) for a,b in v['kernel_ms_per_step'].items()})
p=e.get('projected_strong_scaling_8')
if p:
    for r,v in p['ranks'].items(): print(r, v.get('error') or (round(v['ms_per_step'],3), {a:round(b,3) for a,b in v['kernel_ms_per_step'].items()}, v['real_particles'], v['ghost_particles'], v['bytes_per_face']))
    print(json.dumps({k:v for k,v in p.items() if k not in ('ranks',)}, indent=1))
print(e.get('time_stepping'))
