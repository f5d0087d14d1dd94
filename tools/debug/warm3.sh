run() { python bench.py --no-cpu-baseline --no-extras --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['extra'].get('updates_without_round_trip'))"; }
for i in 1 2; do
echo "default"; run
echo "no check"; run --no-check
echo "dam"; run --workload dam_break
echo "tg"; run --workload taylor_green
done
