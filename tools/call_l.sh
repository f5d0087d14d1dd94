mkdir -p gpurun_out/r04l; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_schedules.py -m gpu -x -q -k "tension or state_fused" 2>&1 | tail -15 > gpurun_out/r04l/t1.log
bash tools/gpu_ab.sh r04l --opts "tension_flag=1;tension_flag=0" -- "rings64_rest|--workload elastic --rings-unperturbed" "rings32_rest|--workload elastic --rings-unperturbed --dtype f32" "rings64|--workload elastic"
cat gpurun_out/r04l/t1.log
