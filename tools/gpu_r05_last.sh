#!/bin/bash
# round 5, the last GPU call: the default bench line on the final tree + kernel stats / step timeline of the two
# configurations whose launches changed after profiles/collect.sh ran (Taylor-Green: periodic images; the cube: unchanged
# kernels, re-taken for the timeline)
mkdir -p gpurun_out/r05z
( time python bench.py > gpurun_out/r05z/bench_default.json 2> gpurun_out/r05z/bench_default.err ) 2> gpurun_out/r05z/bench_default.time
cat gpurun_out/r05z/bench_default.time; tail -c 300 gpurun_out/r05z/bench_default.err
bash tools/prof_one.sh r05z tg --workload taylor_green 2>&1 | tail -45 > gpurun_out/r05z/tg_step_timeline.txt
bash tools/prof_one.sh r05z cube 2>&1 | tail -12 > gpurun_out/r05z/cube_step_timeline.txt
cat gpurun_out/r05z/tg_step_timeline.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05z/bench_default.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'], 3), d['value'], d['kernel_ms_per_step'], 'async', d.get('nnps_updates_without_round_trip'))
r = d['roofline']; print('roofline', r['kernel'][:90], r['frac'], r['traffic'], r['traffic_source'][:40], r['traffic_profiled_kernel_ms'], r['avg_kernel_ms'])
e = d.get('extra', {})
for k, v in e.get('secondary', {}).items():
    print('  ', k[:60], v.get('ms_per_step'), v.get('kernel_ms_per_step'), v.get('parity_max_rel'), v.get('parity_elementwise_max_rel'), v.get('parity_ok'), v.get('error'))
for k, v in e.get('step_vs_n', {}).items():
    print('  step_vs_n', k, v)
p = e.get('projected_strong_scaling_8', {})
print('projection', {k: v for k, v in p.items() if k != 'ranks'})
for k, v in p.get('ranks', {}).items():
    print('   rank', k, v)
print('time_stepping', e.get('time_stepping'))
print('cpu', d.get('cpu_baseline'))
PY
