run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"; }
for i in 1 2 3; do
echo "merged-rsq cube"; run
echo "two-transcendental cube"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run
done
echo "merged dam4m"; run --workload dam_break --dx 0.0055
echo "two dam4m"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run --workload dam_break --dx 0.0055
echo "merged cube.py"; run --params cube
echo "two cube.py"; SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run --params cube
