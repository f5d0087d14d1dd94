#!/bin/bash
# GPU call L: full suite after the block-count selection; Taylor-Green / self-slab timing
mkdir -p gpurun_out/r03l
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03l/pytest.log 2>&1
tail -4 gpurun_out/r03l/pytest.log
python tools/halo_profile.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | head -9 | tee gpurun_out/r03l/halo_cube.log
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
for cfg in "" "--self-slab" "--workload taylor_green" "--workload taylor_green --self-slab"; do
  $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-45s' % '$cfg', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done
