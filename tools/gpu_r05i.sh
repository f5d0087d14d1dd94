#!/bin/bash
mkdir -p gpurun_out/r05i
SPHOPT="--no-cpu-baseline --no-extras --no-counters --no-check --steps 10 --warmup 3"
for ws in 0.75 1.0; do for r in 1 7; do
python bench.py $SPHOPT --workload dam_break --dx 0.0035 --emulate-rank $r/8 --slab-weight-solid $ws > gpurun_out/r05i/rank${r}_$ws.json 2>/dev/null
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05i/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d['config']['workload'][-110:])
    except Exception as e:
        print(f, 'FAILED', e)
PY
