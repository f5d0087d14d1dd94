#!/bin/bash
# GPU call K: the halo tests after the two-pass select+pack / single-round-trip header read; exchange cost
mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_cabi.py tests/test_periodic.py tests/test_hip_parity.py -m gpu -q -x -k "slab or halo or rank or comm or migrat or periodic or cabi or two" ) > gpurun_out/r03k/pytest.log 2>&1
tail -4 gpurun_out/r03k/pytest.log
python tools/halo_profile.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | tee gpurun_out/r03k/halo_cube.log
python tools/halo_profile.py --workload taylor_green 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | tail -9 | tee gpurun_out/r03k/halo_tg.log
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 5"
for cfg in "" "--self-slab" "--workload taylor_green" "--workload taylor_green --self-slab"; do
  $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-45s' % '$cfg', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done
