for ax in 0 1; do for r in 0/8 3/8 7/8; do
python bench.py --workload dam_break --dx 0.0035 --emulate-rank $r --slab-axis $ax --no-cpu-baseline --no-extras --no-counters --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('axis $ax rank $r', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['config']['particles_per_gpu'])"
done; done
