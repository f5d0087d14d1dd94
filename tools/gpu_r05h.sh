#!/bin/bash
mkdir -p gpurun_out/r05h
timeout 1500 python -m pytest tests/test_particle_sort.py tests/test_nnps_reference_cases.py tests/test_bench_multirank.py -q -m gpu -x > gpurun_out/r05h/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05h/tests.log
SPHOPT="--no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3"
for w in 1 2; do
python bench.py $SPHOPT --workload dam_break --dx 0.0035 --emulate-rank 7/8 --opt sort_waves=$w > gpurun_out/r05h/rank7_w$w.json 2>/dev/null
python bench.py $SPHOPT --workload dam_break --dx 0.0055 --opt sort_waves=$w > gpurun_out/r05h/dam4_w$w.json 2>/dev/null
python bench.py $SPHOPT --workload dam_break --opt sort_waves=$w > gpurun_out/r05h/c2_w$w.json 2>/dev/null
python bench.py $SPHOPT --workload dam_break --dx 0.0035 --opt sort_waves=$w > gpurun_out/r05h/dam16_w$w.json 2>/dev/null
python bench.py $SPHOPT --n1 100 --opt sort_waves=$w > gpurun_out/r05h/cube100_w$w.json 2>/dev/null
python bench.py $SPHOPT --opt sort_waves=$w > gpurun_out/r05h/cube_w$w.json 2>/dev/null
python bench.py $SPHOPT --workload taylor_green --opt sort_waves=$w > gpurun_out/r05h/tg_w$w.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05h/*_w*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
