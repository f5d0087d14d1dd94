#!/bin/bash
# One parametrised GPU-box call: A/B timing of library builds x bench configurations, one summary line each.
#   tools/gpu_ab.sh OUT [--libs "a.so b.so"] [--opts "key=val,key=val;key=val"] [--reps N] -- "label|bench args" ...
# --libs  alternative builds of libsphhip.so to swap in (the tree's own build is always the first, "main")
# --opts  sets of library options (bench.py --opt), ';'-separated; every case runs once per set
# Each line: label, ms/step, kernel classes, pair ms per family, parity.  Log: gpurun_out/OUT/ab.log
set -u
OUT=$1; shift
LIBS=""; OPTS=""; REPS=1
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in
    --libs) LIBS="$2"; shift 2;;
    --opts) OPTS="$2"; shift 2;;
    --reps) REPS="$2"; shift 2;;
    *) echo "unknown flag $1"; exit 2;;
  esac
done
shift
mkdir -p gpurun_out/$OUT
export TMPDIR=/tmp
cp pysph_amd/libsphhip.so /tmp/main.so
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
summ() {
  python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('%-40s' % sys.argv[1], 'FAILED'); sys.exit(0)
d=json.loads(ls[-1]); e=d.get('extra',{})
print('%-40s' % sys.argv[1], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()},
      {k: round(v,3) for k,v in d.get('pair_ms_per_family',{}).items()}, 'parity', e.get('parity_max_rel'),
      e.get('parity_elementwise_max_rel'), e.get('parity_neighbour_count_mismatches'))" "$1"
}
{
IFS=';' read -ra OSETS <<< "${OPTS:-;}"
[ ${#OSETS[@]} -eq 0 ] && OSETS=("")
for rep in $(seq $REPS); do
for case in "$@"; do
  label="${case%%|*}"; args="${case#*|}"
  for lib in main $LIBS; do
    if [ $lib = main ]; then cp /tmp/main.so pysph_amd/libsphhip.so; else cp $lib pysph_amd/libsphhip.so; fi
    for os in "${OSETS[@]}"; do
      oa=""; [ -n "$os" ] && for kv in ${os//,/ }; do oa="$oa --opt $kv"; done
      timeout 600 $B $args $oa 2>gpurun_out/$OUT/last.err | summ "$label [$(basename $lib .so)] ${os}"
    done
  done
done
done
cp /tmp/main.so pysph_amd/libsphhip.so
} 2>&1 | tee gpurun_out/$OUT/ab.log
