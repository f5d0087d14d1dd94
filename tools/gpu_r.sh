#!/bin/bash
# GPU call R: phase 1 as resumable loops with one phase-2 call site (no spills in the headline kernel),
# elastic stress sums without the destination's tensors in the loop, occupancy variants
mkdir -p gpurun_out/r03r
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_schedules.py tests/test_baseline_sizes.py -m gpu -x -q ) > gpurun_out/r03r/pytest.log 2>&1
tail -4 gpurun_out/r03r/pytest.log
cp pysph_amd/libsphhip.so /tmp/main.so
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
run() {
  cp $1 pysph_amd/libsphhip.so; shift
  local label="$1"; shift
  $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-50s' % '$label', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, {k: round(v,3) for k,v in d.get('pair_ms_per_family',{}).items()})"
}
{
for L in main alt; do
  F=/tmp/main.so; [ $L = alt ] && F=tools/alt/libsphhip_alt.so
  run $F "$L cube f64"
  run $F "$L cube f32" --dtype f32
  run $F "$L taylor_green" --workload taylor_green
  run $F "$L elastic f64" --workload elastic
  run $F "$L elastic f32" --workload elastic --dtype f32
  run $F "$L dam_break" --workload dam_break
done
cp /tmp/main.so pysph_amd/libsphhip.so
} 2>&1 | tee gpurun_out/r03r/ab.log
