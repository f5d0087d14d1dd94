run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 30 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['nnps'],4))"; }
for i in 1 2; do
for lib in libsphhip libsphhip_nb3072 libsphhip_nb4096; do
echo "$lib cube"; SPH_LIBRARY=$PWD/pysph_amd/$lib.so run
echo "$lib cube100"; SPH_LIBRARY=$PWD/pysph_amd/$lib.so run --n1 100
echo "$lib dam4m"; SPH_LIBRARY=$PWD/pysph_amd/$lib.so run --workload dam_break --dx 0.0055
echo "$lib rank4"; SPH_LIBRARY=$PWD/pysph_amd/$lib.so run --workload dam_break --dx 0.0035 --emulate-rank 4/8
done; done
