#!/bin/bash
mkdir -p gpurun_out/r05n
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "ghost_segments_three_arrays" > gpurun_out/r05n/t.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/r05n/t.log
