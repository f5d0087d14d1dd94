#!/bin/bash
mkdir -p gpurun_out/r05c
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r05c/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r05c/tests.log
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --no-reorder > gpurun_out/r05c/b_unsorted.json 2> gpurun_out/r05c/b_unsorted.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05c/b_unsorted.json').read().strip().splitlines()[-1])
print('unsorted', round(d['ms_per_step'], 3), d['kernel_ms_per_step'])
PY
bash tools/prof_one.sh r05c selfslab --self-slab --n1 142 2>&1 | tail -45
bash tools/prof_one.sh r05c rank7 --workload dam_break --dx 0.0035 --emulate-rank 7/8 2>&1 | tail -30
