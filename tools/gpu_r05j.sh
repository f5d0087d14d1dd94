#!/bin/bash
mkdir -p gpurun_out/r05j
timeout 1500 python -m pytest tests/test_particle_sort.py tests/test_device_helper.py tests/test_nnps_reference_cases.py tests/test_integrator.py -q -m gpu -x > gpurun_out/r05j/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05j/tests.log
SPHOPT="--no-cpu-baseline --no-extras --no-counters --steps 10 --warmup 4"
python bench.py $SPHOPT --no-reorder > gpurun_out/r05j/unsorted.json 2>/dev/null
python bench.py $SPHOPT --no-reorder --opt via_unordered=0 > gpurun_out/r05j/unsorted_novia.json 2>/dev/null
python bench.py $SPHOPT > gpurun_out/r05j/cube.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05j/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d.get('extra', {}).get('parity_max_rel'), d.get('extra', {}).get('parity_neighbour_count_mismatches'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
