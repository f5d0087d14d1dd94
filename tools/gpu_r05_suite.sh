#!/bin/bash
# round 5: the whole GPU suite on the final tree, then the evidence of the round
mkdir -p gpurun_out/r05s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05s/tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05s/tests.log
