run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"; }
for i in 1 2 3; do
echo "run-time exponent, cube"; run
echo "constant 7, cube"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run
done
echo "run-time exponent, dam4m"; run --workload dam_break --dx 0.0055
echo "constant 7, dam4m"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run --workload dam_break --dx 0.0055
echo "run-time exponent, dam4m vh"; run --workload dam_break --dx 0.0055 --vary-h 0.15
echo "constant 7, dam4m vh"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run --workload dam_break --dx 0.0055 --vary-h 0.15
echo "run-time exponent, cube vh"; run --vary-h 0.15
echo "constant 7, cube vh"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run --vary-h 0.15
