#!/bin/bash
# GPU call W: full suite, profiles and the driver's default bench invocation at the final commit
mkdir -p gpurun_out/r03w
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03w/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r03w/pytest.log
bash profiles/collect.sh r03 > gpurun_out/r03w/collect.log 2>&1
( time python bench.py ) > gpurun_out/r03w/bench_default.json 2> gpurun_out/r03w/bench_default.err
tail -3 gpurun_out/r03w/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03w/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])
for k,v in d['extra']['secondary'].items():
    print(k, {kk: v[kk] for kk in ('ms_per_step','parity_ok','parity_max_rel') if kk in v})
PY
