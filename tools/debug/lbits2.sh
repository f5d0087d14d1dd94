run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 30 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['nnps'],4))"; }
for lb in 0 9 10 11; do echo "cube lbits $lb"; run --opt sort_lbits=$lb; done
for lb in 0 9 10 11; do echo "cube100 lbits $lb"; run --n1 100 --opt sort_lbits=$lb; done
for lb in 0 9 10 11; do echo "dam4m lbits $lb"; run --workload dam_break --dx 0.0055 --opt sort_lbits=$lb; done
for lb in 0 9 10 11; do echo "C2 lbits $lb"; run --workload dam_break --opt sort_lbits=$lb; done
for lb in 0 9 10 11; do echo "rings lbits $lb"; run --workload elastic --opt sort_lbits=$lb; done
for lb in 0 9 10 11; do echo "C4 lbits $lb"; run --workload dam_break --dx 0.0035 --steps 10 --opt sort_lbits=$lb; done
