mkdir -p gpurun_out/r04g; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_nnps_reference_cases.py tests/test_hip_parity.py tests/test_periodic.py tests/test_device_helper.py tests/test_cabi.py tests/test_ghost_sets.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04g/t1.log
bash tools/gpu_ab.sh r04g -- "cube|" "cube_unsorted|--no-reorder" "C2|--workload dam_break --dx 0.0087" "dam4M|--workload dam_break --dx 0.0055" "dam16M|--workload dam_break --dx 0.0035" "TG|--workload taylor_green" "rings|--workload elastic"
cat gpurun_out/r04g/t1.log
