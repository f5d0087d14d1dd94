#!/bin/bash
# GPU call X: uniform-mass records (p / rho^2 in the mass slot, most of the per-record EOS gone)
mkdir -p gpurun_out/r03x
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_schedules.py tests/test_hip_parity.py -m gpu -x -q ) > gpurun_out/r03x/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r03x/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run() {
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-32s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'parity', d.get('extra',{}).get('parity_max_rel'), d.get('extra',{}).get('parity_neighbour_count_mismatches'))"
}
{
for m in 0 1 0 1; do run "cube f64 mass_fuse=$m" --opt mass_fuse=$m; done
for m in 0 1; do
  run "cube f32 mass_fuse=$m" --dtype f32 --opt mass_fuse=$m
  run "dam_break mass_fuse=$m" --workload dam_break --opt mass_fuse=$m
  run "dam_break 0.0055 mass_fuse=$m" --workload dam_break --dx 0.0055 --opt mass_fuse=$m --no-check
done
} 2>&1 | tee gpurun_out/r03x/ab.log
