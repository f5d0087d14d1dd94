#!/bin/bash
# GPU call F: packed fp32 body -- parity, then A/B against the scalar body (--ablate 9 keeps the scalar one)
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_schedules.py tests/test_baseline_sizes.py -m gpu -q -k "fp32 or f32 or eos" ) > gpurun_out/r03f/pytest.log 2>&1
tail -6 gpurun_out/r03f/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --dtype f32"
for cfg in "" "--ablate 9"; do
  $B $cfg > gpurun_out/r03f/f32_$(echo $cfg | tr ' -' '__').json 2>/dev/null
  python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r03f/f32_$(echo $cfg | tr ' -' '__').json') if l.startswith('{')][-1])
print('f32 $cfg', d['ms_per_step'], d['kernel_ms_per_step'], d.get('extra',{}).get('parity_max_rel'), d.get('extra',{}).get('parity_neighbour_count_mismatches'))
P
done
