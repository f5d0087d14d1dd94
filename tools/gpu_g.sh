#!/bin/bash
# GPU call G: record-line prefetch in phase 1, A/B over the workloads
mkdir -p gpurun_out/r03g
export TMPDIR=/tmp
timeout 1500 python tools/ab_r03.py > gpurun_out/r03g/ab.log 2>&1
grep -v "^{" gpurun_out/r03g/ab.log | tail -20
grep error gpurun_out/r03g/ab.log | head -3
