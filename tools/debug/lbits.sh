run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 30 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"; }
for r in 0 4; do
for lb in 0 9 10 11; do echo "rank $r lbits $lb"; run --workload dam_break --dx 0.0035 --emulate-rank $r/8 --opt sort_lbits=$lb; done
done
