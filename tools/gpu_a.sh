#!/bin/bash
# GPU call A of round 3: the whole -m gpu suite, the default bench line, the A/B table
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03a/pytest.log 2>&1
tail -5 gpurun_out/r03a/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err
tail -c 600 gpurun_out/r03a/bench_default.err
timeout 1200 python tools/ab_r03.py > gpurun_out/r03a/ab.log 2>&1
tail -25 gpurun_out/r03a/ab.log
timeout 600 python tools/count_iters.py > gpurun_out/r03a/iters.log 2>&1
tail -8 gpurun_out/r03a/iters.log
