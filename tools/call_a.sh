mkdir -p gpurun_out/r04a; export TMPDIR=/tmp
python -m pytest tests/test_cabi.py -m gpu -x -q -k "fixed_h" 2>&1 | tail -3 > gpurun_out/r04a/t1.log
timeout 900 python -m pytest tests/test_baseline_sizes.py -m gpu -x -q -k "16m or taylor or rings_2m_vs" 2>&1 | tail -15 > gpurun_out/r04a/t2.log
python bench.py --workload dam_break --dx 0.0035 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/r04a/dam16m.json 2> gpurun_out/r04a/dam16m.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04a/prof -- python $GRAFT_REPO_ROOT/bench.py --workload dam_break --dx 0.0035 --no-cpu-baseline --no-extras --no-check --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r04a/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04a/dam16m_kernel_stats.csv \; ; rm -rf gpurun_out/r04a/prof
cat gpurun_out/r04a/t1.log gpurun_out/r04a/t2.log; tail -c 1500 gpurun_out/r04a/dam16m.json
