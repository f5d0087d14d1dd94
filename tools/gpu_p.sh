#!/bin/bash
# GPU call P: wavefronts per workgroup (x-adjacent tiles) with the mod-3 row order
mkdir -p gpurun_out/r03p
export TMPDIR=/tmp
cp pysph_amd/libsphhip.so /tmp/libsphhip_wpb1.so
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  cp $1 pysph_amd/libsphhip.so; shift
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-50s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
}
{
for W in 1 2 4; do
  L=tools/alt/libsphhip_wpb$W.so; [ $W = 1 ] && L=/tmp/libsphhip_wpb1.so
  run $L "wpb$W cube f64"
  run $L "wpb$W cube f32" --dtype f32
  run $L "wpb$W taylor_green" --workload taylor_green
  run $L "wpb$W elastic" --workload elastic
  run $L "wpb$W dam_break" --workload dam_break
done
cp /tmp/libsphhip_wpb1.so pysph_amd/libsphhip.so
} 2>&1 | tee gpurun_out/r03p/ab.log
