#!/bin/bash
# GPU call S: one gathered record in flight ahead (phase 2 pipelined by one) at full occupancy, no spills
mkdir -p gpurun_out/r03s
export TMPDIR=/tmp
cp pysph_amd/libsphhip.so /tmp/main.so
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run() {
  cp $1 pysph_amd/libsphhip.so; shift
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'parity', d.get('extra',{}).get('parity_max_rel'), d.get('extra',{}).get('parity_neighbour_count_mismatches'))"
}
{
for L in main alt main alt; do
  F=/tmp/main.so; [ $L = alt ] && F=tools/alt/libsphhip_alt.so
  run $F "$L cube f64"
done
for L in main alt; do
  F=/tmp/main.so; [ $L = alt ] && F=tools/alt/libsphhip_alt.so
  run $F "$L cube f32" --dtype f32
  run $F "$L dam_break" --workload dam_break
  run $F "$L dam_break 0.0055" --workload dam_break --dx 0.0055 --no-check
done
cp /tmp/main.so pysph_amd/libsphhip.so
} 2>&1 | tee gpurun_out/r03s/ab.log
