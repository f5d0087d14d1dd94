#!/bin/bash
# GPU call U: self-exchange slab (RCCL halo path on one GPU) at HEAD
mkdir -p gpurun_out/r03u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
for cfg in "" "--self-slab" "--workload taylor_green" "--workload taylor_green --self-slab" "--fixed-bounds" "--fixed-bounds --self-slab"; do
  $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-45s' % '$cfg', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done 2>&1 | tee gpurun_out/r03u/self_slab.log
python tools/halo_profile.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | head -12 | tee gpurun_out/r03u/halo_cube.log
