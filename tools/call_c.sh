mkdir -p gpurun_out/r04c; export TMPDIR=/tmp
python -m pytest tests/test_cabi.py -m gpu -x -q -k "fixed_h" 2>&1 | tail -5 > gpurun_out/r04c/t1.log
timeout 900 python -m pytest tests/test_schedules.py -m gpu -x -q -k "state_fused" 2>&1 | tail -30 > gpurun_out/r04c/t2.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fused_records" 2>&1 | tail -30 > gpurun_out/r04c/t3.log
bash tools/gpu_ab.sh r04c --opts "mass_fuse=1;mass_fuse=0" -- "TG|--workload taylor_green" "rings64|--workload elastic" "rings32|--workload elastic --dtype f32" 
cat gpurun_out/r04c/t1.log gpurun_out/r04c/t2.log gpurun_out/r04c/t3.log
