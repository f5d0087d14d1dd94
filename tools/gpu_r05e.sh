#!/bin/bash
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_particle_sort.py tests/test_nnps_reference_cases.py -q -m gpu -x > gpurun_out/r05e/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05e/tests.log
for ws in 0.4; do for r in 0 7; do
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0035 --emulate-rank $r/8 --slab-weight-solid $ws > gpurun_out/r05e/rank${r}_$ws.json 2> gpurun_out/r05e/rank${r}_$ws.err
done; done
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0035 > gpurun_out/r05e/dam16.json 2> gpurun_out/r05e/dam16.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 20 --warmup 5 > gpurun_out/r05e/cube.json 2> gpurun_out/r05e/cube.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0055 > gpurun_out/r05e/dam4.json 2> gpurun_out/r05e/dam4.err
python bench.py --no-cpu-baseline --no-extras --no-counters --steps 10 --warmup 3 --workload dam_break > gpurun_out/r05e/c2.json 2> gpurun_out/r05e/c2.err
python - <<'PY'
import json
for n in ('rank0_0.4', 'rank7_0.4', 'dam16', 'cube', 'dam4', 'c2'):
    try:
        d = json.loads(open('gpurun_out/r05e/%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d['config']['particles_per_gpu'], d['config']['workload'][-120:])
    except Exception as e:
        print(n, 'FAILED', e); print(open('gpurun_out/r05e/%s.err' % n).read()[-800:])
PY
bash tools/prof_one.sh r05e rank7 --workload dam_break --dx 0.0035 --emulate-rank 7/8 2>&1 | tail -12
