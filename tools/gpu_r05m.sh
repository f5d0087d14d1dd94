#!/bin/bash
# round 5: sph_halo_select_pack in two launches -- the tests that cross it, then the slab's cost at a rank's size
mkdir -p gpurun_out/r05m
timeout 1200 python -m pytest tests/test_cabi.py tests/test_bench_multirank.py tests/test_schedules.py tests/test_hip_parity.py tests/test_integrator.py -q -m gpu -x -k "cabi or select_pack or multirank or bench or padded or slab or halo or rank or rccl or overlap or exchange" > gpurun_out/r05m/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05m/tests.log
SPHOPT="--no-cpu-baseline --no-extras --no-counters --steps 20 --warmup 5 --n1 142"
python bench.py $SPHOPT > gpurun_out/r05m/plain142.json 2>/dev/null
python bench.py $SPHOPT --self-slab > gpurun_out/r05m/selfslab142.json 2>gpurun_out/r05m/selfslab142.err
python tools/halo_profile.py --n1 142 > gpurun_out/r05m/halo_profile.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05m/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d.get('extra', {}).get('parity_max_rel'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
tail -25 gpurun_out/r05m/halo_profile.txt
