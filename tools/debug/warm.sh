run() { python bench.py --no-cpu-baseline --no-extras --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['pair'],4))"; }
for i in 1 2 3; do
echo "warmup 5"; run --steps 20 --warmup 5
echo "warmup 60"; run --steps 20 --warmup 60
echo "warmup 5, no parity check"; run --steps 20 --warmup 5 --no-check
done
