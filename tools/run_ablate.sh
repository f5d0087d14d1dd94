#!/bin/bash
# run a list of bench configurations on the GPU box, one JSON line per config
mkdir -p gpurun_out/abl
for cfg in "$@"; do
    name=$(echo "$cfg" | tr ' =-' '___')
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $cfg > gpurun_out/abl/$name.json 2> gpurun_out/abl/$name.err
    python - <<P
import json
try:
    d=json.load(open('gpurun_out/abl/$name.json'))
    print('$cfg', 'ms/step %.3f' % d['ms_per_step'], 'pair %.3f' % d['kernel_ms_per_step']['pair'], d['kernel_ms_per_step'])
except Exception as e:
    print('$cfg', 'FAILED', e)
P
done
