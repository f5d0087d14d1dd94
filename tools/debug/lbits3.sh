run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 30 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['nnps'],4))"; }
echo cube; run
echo cube100; run --n1 100
echo cube63; run --n1 63
echo dam4m; run --workload dam_break --dx 0.0055
echo C2; run --workload dam_break
echo rings; run --workload elastic
echo tg; run --workload taylor_green
echo C4; run --workload dam_break --dx 0.0035 --steps 10
echo rank0; run --workload dam_break --dx 0.0035 --emulate-rank 0/8
echo rank4; run --workload dam_break --dx 0.0035 --emulate-rank 4/8
echo rank7; run --workload dam_break --dx 0.0035 --emulate-rank 7/8
echo cubevh; run --vary-h 0.15
echo unsorted; run --no-reorder
echo cube252; run --n1 252 --steps 10
