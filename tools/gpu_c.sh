#!/bin/bash
# GPU call C of round 3: full suite after the halo / CSR / nnps changes; halo overhead; CSR timing
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03c/pytest.log 2>&1
tail -5 gpurun_out/r03c/pytest.log
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
for cfg in "" "--self-slab" "--fixed-bounds" "--self-slab --fixed-bounds" \
           "--workload taylor_green" "--workload taylor_green --self-slab" "--workload taylor_green --fixed-bounds"; do
  name=$(echo "cube $cfg" | tr ' =-' '___')
  $B $cfg > gpurun_out/r03c/$name.json 2> gpurun_out/r03c/$name.err
  python - <<P
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03c/$name.json') if l.startswith('{')][-1])
    print('%-50s ms/step %.3f  %s' % ('$cfg', d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}))
except Exception as e:
    print('$cfg', 'FAILED', e, open('gpurun_out/r03c/$name.err').read()[-600:])
P
done
timeout 600 python tools/csr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03c/csr.log
