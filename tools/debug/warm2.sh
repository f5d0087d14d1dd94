run() { python bench.py --no-cpu-baseline --no-extras --no-counters "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['pair'],4))"; }
for i in 1 2; do
for w in 5 11 20 40 100; do echo "warmup $w"; run --steps 20 --warmup $w; done
done
