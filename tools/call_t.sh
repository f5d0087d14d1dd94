mkdir -p gpurun_out/r04t; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_kernel_moments.py tests/test_oracle_golden.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04t/t1.log
timeout 900 python -m pytest tests/test_baseline_sizes.py -m gpu -x -q -k "taylor or rings or elastic" 2>&1 | tail -6 >> gpurun_out/r04t/t1.log
bash tools/gpu_ab.sh r04t --reps 2 -- "TG|--workload taylor_green" "rings64|--workload elastic" "rings32|--workload elastic --dtype f32" "rings32_rest|--workload elastic --dtype f32 --rings-unperturbed"
cat gpurun_out/r04t/t1.log
