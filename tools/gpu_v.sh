#!/bin/bash
# GPU call V: pair kernel without its profiling hooks (scalar registers), EOS flags as a compile-time constant
mkdir -p gpurun_out/r03v
export TMPDIR=/tmp
cp pysph_amd/libsphhip.so /tmp/main.so
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run() {
  cp $1 pysph_amd/libsphhip.so; shift
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-28s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, {k: round(v,3) for k,v in d.get('pair_ms_per_family',{}).items()}, 'parity', d.get('extra',{}).get('parity_max_rel'), d.get('extra',{}).get('parity_neighbour_count_mismatches'))"
}
{
for L in old new old new; do
  F=/tmp/main.so; [ $L = old ] && F=tools/alt/libsphhip_old.so
  run $F "$L cube f64"
done
for L in old new; do
  F=/tmp/main.so; [ $L = old ] && F=tools/alt/libsphhip_old.so
  run $F "$L cube f32" --dtype f32
  run $F "$L taylor_green" --workload taylor_green --no-check
  run $F "$L elastic f64" --workload elastic --no-check
  run $F "$L elastic f32" --workload elastic --dtype f32 --no-check
  run $F "$L dam_break" --workload dam_break
done
cp /tmp/main.so pysph_amd/libsphhip.so
} 2>&1 | tee gpurun_out/r03v/ab.log
