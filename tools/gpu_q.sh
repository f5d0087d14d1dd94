#!/bin/bash
# GPU call Q: loop nesting of the row steps, traversal block rows for the wide-record workloads
mkdir -p gpurun_out/r03q
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-50s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
}
{
for wl in cube taylor_green elastic; do
  for m in 3 11; do run "$wl row_mod3=$m" --workload $wl --opt row_mod3=$m; done
done
for wl in taylor_green elastic dam_break; do
  for by in 4 6 12; do run "$wl tile_block_rows=$by" --workload $wl --opt tile_block_rows=$by; done
done
} 2>&1 | tee gpurun_out/r03q/ab.log
