# A/B of the row-LDS pair kernel (option row_lds) on the S-rings (C5), fp64 and fp32; the first run of each keeps bench.py's parity check
run() { python bench.py --no-cpu-baseline --no-extras --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d.get('extra',{}).get('parity_check'))"; }
for dt in f64 f32; do
echo "wave $dt"; run --workload elastic --dtype $dt
echo "rowlds $dt"; run --workload elastic --dtype $dt --opt row_lds=1
echo "wave $dt"; run --workload elastic --dtype $dt --no-check
echo "rowlds $dt"; run --workload elastic --dtype $dt --opt row_lds=1 --no-check
done
