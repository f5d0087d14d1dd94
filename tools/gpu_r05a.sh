#!/bin/bash
# round 5, call A: the new particle sort / round-trip-free update -- targeted tests, then timing + kernel trace
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_particle_sort.py tests/test_nnps_reference_cases.py tests/test_cabi.py -x -q -m gpu > gpurun_out/r05a/t1.log 2>&1
echo "t1 rc=$?"; tail -15 gpurun_out/r05a/t1.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_periodic.py tests/test_ghost_sets.py -x -q -m gpu > gpurun_out/r05a/t2.log 2>&1
echo "t2 rc=$?"; tail -15 gpurun_out/r05a/t2.log
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/r05a/b_cube.json 2> gpurun_out/r05a/b_cube.err; tail -c 1500 gpurun_out/r05a/b_cube.json
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --opt async_update=0 > gpurun_out/r05a/b_cube_sync.json 2> gpurun_out/r05a/b_cube_sync.err
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --n1 100 > gpurun_out/r05a/b_cube100.json 2> gpurun_out/r05a/b_cube100.err
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --workload dam_break > gpurun_out/r05a/b_dam.json 2> gpurun_out/r05a/b_dam.err
python - <<'PY'
import json
for n in ('b_cube', 'b_cube_sync', 'b_cube100', 'b_dam'):
    try:
        d = json.loads(open('gpurun_out/r05a/%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d.get('extra', {}).get('parity_max_rel'), d.get('extra', {}).get('parity_neighbour_count_mismatches'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
bash tools/prof_one.sh r05a cube 2>&1 | tail -60
bash tools/prof_one.sh r05a dam --workload dam_break 2>&1 | tail -45
