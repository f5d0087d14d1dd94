mkdir -p gpurun_out/r04f; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_schedules.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04f/t1.log
bash tools/gpu_ab.sh r04f --opts "mass_fuse=1;mass_fuse=0" -- "cube_vh|--vary-h 0.15" "cube_vh_f32|--vary-h 0.15 --dtype f32" "cube|"
cat gpurun_out/r04f/t1.log
