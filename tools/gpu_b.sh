#!/bin/bash
# GPU call B of round 3: new schedule tests, nnps timing, pipelined-kernel sweep
mkdir -p gpurun_out/r03b
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_schedules.py tests/test_hip_parity.py -m gpu -x -q ) > gpurun_out/r03b/pytest.log 2>&1
tail -5 gpurun_out/r03b/pytest.log
timeout 1500 python tools/ab_r03.py > gpurun_out/r03b/ab.log 2>&1
grep -v "^{" gpurun_out/r03b/ab.log | tail -45
