mkdir -p gpurun_out/r04u; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_multirank.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04u/t1.log
bash tools/gpu_ab.sh r04u --reps 2 -- "cube|" "selfslab_plain|--self-slab" "selfslab_overlap_prepare|--self-slab --overlap-halo" "selfslab_overlap_splitpair|--self-slab --overlap-halo --opt split_pair=1"
cat gpurun_out/r04u/t1.log
