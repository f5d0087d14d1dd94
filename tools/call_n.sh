mkdir -p gpurun_out/r04n; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_multirank.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r04n/t1.log
bash tools/gpu_ab.sh r04n --reps 2 -- "cube|" "selfslab_overlap|--self-slab --overlap-halo" "selfslab_plain|--self-slab"
cat gpurun_out/r04n/t1.log
