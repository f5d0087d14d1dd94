#!/bin/bash
mkdir -p gpurun_out/r05k
SPH_FUZZ_SEEDS=48 timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "randomised" > gpurun_out/r05k/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -6 gpurun_out/r05k/fuzz.log
