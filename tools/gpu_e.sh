#!/bin/bash
# GPU call E of round 3: the whole -m gpu suite, the rocprofv3 evidence (profiles/collect.sh r03), the default bench line
mkdir -p gpurun_out/r03e
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03e/pytest.log 2>&1
tail -5 gpurun_out/r03e/pytest.log
( time bash profiles/collect.sh r03 ) > gpurun_out/r03e/collect.log 2>&1
tail -3 gpurun_out/r03e/collect.log
( time timeout 900 python bench.py ) > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err
tail -c 300 gpurun_out/r03e/bench_default.err
for o in 4 16; do
  python bench.py --no-cpu-baseline --no-check --no-extras --steps 10 --warmup 3 --opt tile_block_rows=$o > gpurun_out/r03e/tbr_$o.json 2>/dev/null
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03e/tbr_$o.json') if l.startswith('{')][-1]); print('tile_block_rows=$o', d['ms_per_step'], d['kernel_ms_per_step'])"
done
