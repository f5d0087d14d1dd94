#!/bin/bash
# kernel-trace stats of one bench configuration: tools/prof_one.sh OUTDIR NAME bench-args...
OUT=$1; NAME=$2; shift 2
ROOT=$(pwd); mkdir -p gpurun_out/$OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o run -- python $ROOT/bench.py --no-cpu-baseline --no-check --no-extras --no-counters --steps 10 --warmup 3 "$@" > $ROOT/gpurun_out/$OUT/$NAME.log 2>&1
find /tmp/prof_$NAME -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/$OUT/${NAME}_kernel_stats.csv \;
cd $ROOT
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$OUT/${NAME}_kernel_stats.csv')))
print('== $NAME')
for r in rows[:22]:
    print('%-100s %6s %10.1f %9.1f' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
# per-launch durations of the pair kernels, in launch order (the last 12)
python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_$NAME/**/*kernel_trace.csv', recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if 'k_pair_wave' in r['Kernel_Name'] and 'FamNbr' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    print('pair launches (us):', [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[-12:]])
    allr = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
    # one steady-state step: kernels between the last two k_bin_keys launches
    idx = [i for i, r in enumerate(allr) if 'k_bin_keys' in r['Kernel_Name']]
    if len(idx) >= 3:
        a, b = idx[-3], idx[-2]
        t0 = int(allr[a]['Start_Timestamp'])
        for r in allr[a:b]:
            print('%9.1f %9.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:90]))
PY
