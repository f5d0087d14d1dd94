#!/bin/bash
# GPU call T: full suite at HEAD, occupancy sweep of the final kernel, the driver's default bench invocation
mkdir -p gpurun_out/r03t
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03t/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r03t/pytest.log
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
for pad in 0 1200 3100 5800; do
  $B --opt lds_pad=$pad 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lds_pad=$pad', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done 2>&1 | tee gpurun_out/r03t/occupancy.log
( time python bench.py ) > gpurun_out/r03t/bench_default.json 2> gpurun_out/r03t/bench_default.err
tail -3 gpurun_out/r03t/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03t/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])
for k,v in d['extra'].get('secondary',{}).items() if isinstance(d['extra'].get('secondary'),dict) else enumerate(d['extra'].get('secondary',[])):
    print(k, {kk: v[kk] for kk in ('ms_per_step','value','parity_ok','parity_max_rel') if kk in v})
PY
