#!/bin/bash
# GPU call Y: uniform-mass records, mass range folded into the final reduction kernel
mkdir -p gpurun_out/r03y
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_schedules.py tests/test_cabi.py tests/test_nnps_reference_cases.py -m gpu -x -q ) > gpurun_out/r03y/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r03y/pytest.log
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-32s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
}
{
for m in 0 1 0 1; do run "cube f64 mass_fuse=$m" --opt mass_fuse=$m; done
for m in 0 1; do
  run "cube f32 mass_fuse=$m" --dtype f32 --opt mass_fuse=$m
  run "dam_break mass_fuse=$m" --workload dam_break --opt mass_fuse=$m
  run "cube.py params mass_fuse=$m" --params cube --opt mass_fuse=$m
  run "unsorted mass_fuse=$m" --no-reorder --opt mass_fuse=$m
done
} 2>&1 | tee gpurun_out/r03y/ab.log
