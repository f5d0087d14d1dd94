mkdir -p gpurun_out/r04i; export TMPDIR=/tmp
bash tools/gpu_ab.sh r04i --reps 2 -- "cube|" "cube_f32|--dtype f32" "dam4M|--workload dam_break --dx 0.0055" "cube_vh|--vary-h 0.15" "cube.py|--params cube"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04i/t_all.log
cat gpurun_out/r04i/t_all.log
