# headline, 100^3, C2 and the 4 M dam break without the extras: ms per step and the kernel classes
run() { python bench.py --no-cpu-baseline --no-extras --no-counters --no-check "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"; }
for i in 1 2; do
echo "cube 159"; run
echo "cube 100"; run --n1 100
echo "C2"; run --workload dam_break
echo "dam4m"; run --workload dam_break --dx 0.0055
echo "dam4m vh"; run --workload dam_break --dx 0.0055 --vary-h 0.15
done
