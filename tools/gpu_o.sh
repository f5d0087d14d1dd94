#!/bin/bash
# GPU call O: non-temporal stores of the pair kernel's results (L2 capacity for the gathered rows)
mkdir -p gpurun_out/r03o
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-50s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
}
{
for m in 0 1 0 1; do run "cube f64 nt_out=$m" --opt nt_out=$m; done
for m in 0 1; do run "cube f32 nt_out=$m" --dtype f32 --opt nt_out=$m; done
for wl in taylor_green elastic dam_break; do
  for m in 0 1; do run "$wl nt_out=$m" --workload $wl --opt nt_out=$m; done
done
} 2>&1 | tee gpurun_out/r03o/ab.log
