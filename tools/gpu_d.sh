#!/bin/bash
# GPU call D of round 3: variant 7 (LDS-resident workgroup tiles): correctness, then timing
mkdir -p gpurun_out/r03d
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_schedules.py -m gpu -q -k "lds_resident" ) > gpurun_out/r03d/pytest.log 2>&1
tail -12 gpurun_out/r03d/pytest.log
timeout 1500 python tools/ab_r03.py "lds f" "cube f" > gpurun_out/r03d/ab.log 2>&1
grep -v "^{" gpurun_out/r03d/ab.log | tail -30
grep error gpurun_out/r03d/ab.log | head -5
