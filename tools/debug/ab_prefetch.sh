run() { python bench.py --no-cpu-baseline --no-extras --no-counters --steps 20 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['pair'],4), d['extra'].get('parity_ok'), d['extra'].get('parity_max_rel'))"; }
for w in "" "--n1 100" "--workload dam_break --dx 0.0055" "--workload taylor_green" "--workload elastic" "--vary-h 0.15" "--dtype f32"; do
echo "prefetch $w: $(run $w)"
echo "before   $w: $(SPH_LIBRARY=$PWD/pysph_amd/libsphhip_ab.so run $w)"
done
