#!/bin/bash
# the default bench line as the driver runs it, on the final tree
mkdir -p gpurun_out/r05
( time python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err ) 2> gpurun_out/r05/bench_default.time
cat gpurun_out/r05/bench_default.time; tail -c 300 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'], 3), d['value'], d['kernel_ms_per_step'], 'async', d.get('nnps_updates_without_round_trip'))
r = d['roofline']; print('roofline', r['frac'], r['traffic'], r['traffic_source'][:40], r['traffic_profiled_kernel_ms'], r['avg_kernel_ms'])
p = d['extra'].get('projected_strong_scaling_8', {})
print('projection', {k: v for k, v in p.items() if k != 'ranks'})
for k, v in p.get('ranks', {}).items():
    print('   rank', k, v['ms_per_step'], v['kernel_ms_per_step'], v['real_per_array'])
PY
