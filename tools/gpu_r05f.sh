#!/bin/bash
mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests/test_particle_sort.py tests/test_schedules.py -q -m gpu -x -k "sort or merged or stable or dense or several or round_trip or async" > gpurun_out/r05f/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r05f/tests.log
timeout 900 python -m pytest tests/test_baseline_sizes.py -q -m gpu -x -k "variable_h_runs_merged" > gpurun_out/r05f/tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 gpurun_out/r05f/tests2.log
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0055 --vary-h 0.15 > gpurun_out/r05f/dam4vh.json 2> gpurun_out/r05f/dam4vh.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0055 --vary-h 0.15 --opt merge_arrays=0 > gpurun_out/r05f/dam4vh_nomerge.json 2> gpurun_out/r05f/dam4vh_nomerge.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0055 > gpurun_out/r05f/dam4.json 2> gpurun_out/r05f/dam4.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0035 --emulate-rank 7/8 > gpurun_out/r05f/rank7.json 2> gpurun_out/r05f/rank7.err
python - <<'PY'
import json
for n in ('dam4vh', 'dam4vh_nomerge', 'dam4', 'rank7'):
    try:
        d = json.loads(open('gpurun_out/r05f/%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(n, 'FAILED', e); print(open('gpurun_out/r05f/%s.err' % n).read()[-800:])
PY
