#!/bin/bash
# round 5: the float builds of generated families (option arith_f32) + the generated-family and C-ABI suites around them
mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cabi.py -q -m gpu -s -k "fp32_arithmetic or cabi or layout" > gpurun_out/r05l/gen.log 2>&1
echo "rc=$?"; grep -i "arith_f32\|passed\|failed\|error" gpurun_out/r05l/gen.log | tail -30
