#!/bin/bash
# kernel-trace stats of one bench configuration: tools/prof_one.sh OUTDIR NAME bench-args...
OUT=$1; NAME=$2; shift 2
ROOT=$(pwd); mkdir -p gpurun_out/$OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o run -- python $ROOT/bench.py --no-cpu-baseline --no-check --no-extras --steps 10 --warmup 3 "$@" > $ROOT/gpurun_out/$OUT/$NAME.log 2>&1
find /tmp/prof_$NAME -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/$OUT/${NAME}_kernel_stats.csv \;
cd $ROOT
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$OUT/${NAME}_kernel_stats.csv')))
print('== $NAME')
for r in rows[:22]:
    print('%-100s %6s %10.1f %9.1f' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
