#!/bin/bash
# GPU call N: order in which a wavefront visits its 3x3 rows of cells (option row_mod3), x traversal block rows
mkdir -p gpurun_out/r03n
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-50s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
}
{
for m in 0 1 2 3 4; do run "cube f64 row_mod3=$m" --opt row_mod3=$m; done
for by in 3 6 9 12 16; do run "cube f64 row_mod3=3 tile_block_rows=$by" --opt row_mod3=3 --opt tile_block_rows=$by; done
run "cube f64 row_mod3=0 tile_block_rows=6" --opt tile_block_rows=6
for m in 0 3; do run "cube f32 row_mod3=$m" --dtype f32 --opt row_mod3=$m; done
for wl in taylor_green elastic dam_break; do
  for m in 0 3; do run "$wl row_mod3=$m" --workload $wl --opt row_mod3=$m; done
done
for m in 0 3; do run "dam_break 0.0055 row_mod3=$m" --workload dam_break --dx 0.0055 --opt row_mod3=$m; done
} 2>&1 | tee gpurun_out/r03n/ab.log
