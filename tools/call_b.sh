mkdir -p gpurun_out/r04b; export TMPDIR=/tmp
python -m pytest tests/test_cabi.py -m gpu -x -q -k "fixed_h" 2>&1 | tail -40 > gpurun_out/r04b/t1.log
timeout 900 python -m pytest tests/test_schedules.py -m gpu -x -q -k "merged or eos_fused" 2>&1 | tail -40 > gpurun_out/r04b/t2.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04b/t3.log
bash tools/gpu_ab.sh r04b --opts "merge_arrays=1;merge_arrays=0" -- "C2|--workload dam_break --dx 0.0087" "dam4M|--workload dam_break --dx 0.0055" "dam16M|--workload dam_break --dx 0.0035"
cat gpurun_out/r04b/t1.log gpurun_out/r04b/t2.log gpurun_out/r04b/t3.log
