#!/bin/bash
# GPU call H: fp32 TVF / elastic at 4 wavefronts per SIMD; the time-stepping secondary
mkdir -p gpurun_out/r03h
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --dtype f32"
for cfg in "--workload elastic" "--workload taylor_green" "--workload elastic_block --n1 126"; do
  name=$(echo "$cfg" | tr ' -' '__')
  $B $cfg > gpurun_out/r03h/$name.json 2>gpurun_out/r03h/$name.err
  python - <<P
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03h/$name.json') if l.startswith('{')][-1])
    print('f32 $cfg', round(d['ms_per_step'],3), d.get('pair_ms_per_family'), d.get('extra',{}).get('parity_max_rel'), d.get('extra',{}).get('parity_neighbour_count_mismatches'))
except Exception as e:
    print('$cfg FAILED', e, open('gpurun_out/r03h/$name.err').read()[-500:])
P
done
python - <<'P'
import sys, json, torch
sys.path.insert(0, '.')
import bench
t = torch.cuda.Stream(); torch.cuda.set_stream(t)
print(json.dumps(bench.time_stepping(0, t)))
P
( timeout 600 python -m pytest tests/test_baseline_sizes.py tests/test_hip_parity.py -m gpu -q -x -k "fp32 or f32" ) 2>&1 | tail -3
