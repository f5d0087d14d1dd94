#!/bin/bash
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_bench_multirank.py tests/test_particle_sort.py tests/test_cabi.py tests/test_parallel_gloo.py -q -m gpu -x > gpurun_out/r05d/tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05d/tests.log
for r in 0 3 7; do
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --workload dam_break --dx 0.0035 --emulate-rank $r/8 --slab-weight-solid 0.35 > gpurun_out/r05d/rank$r.json 2> gpurun_out/r05d/rank$r.err
done
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --n1 142 > gpurun_out/r05d/plain142.json 2> gpurun_out/r05d/plain142.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --n1 142 --self-slab > gpurun_out/r05d/slab142_padded.json 2> gpurun_out/r05d/slab142_padded.err
python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3 --n1 142 --self-slab --halo-protocol capacity > gpurun_out/r05d/slab142_capacity.json 2> gpurun_out/r05d/slab142_capacity.err
python - <<'PY'
import json
for n in ('rank0', 'rank3', 'rank7', 'plain142', 'slab142_padded', 'slab142_capacity'):
    try:
        d = json.loads(open('gpurun_out/r05d/%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d['config']['particles_per_gpu'], d.get('nnps_updates_without_round_trip'))
    except Exception as e:
        print(n, 'FAILED', e); print(open('gpurun_out/r05d/%s.err' % n).read()[-800:])
PY
bash tools/prof_one.sh r05d selfslab_padded --self-slab --n1 142 2>&1 | tail -22

