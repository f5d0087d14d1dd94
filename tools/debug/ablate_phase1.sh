# phase-1 shares of the headline pair kernel (profiling build: SPH_LIBRARY=.../libsphhip_prof.so): 0 = full, 6 = prologue + finish only,
# 7 = + staging of the row tiles, 2 = + mask arithmetic (phase 1 complete, phase 2 skipped), 3 = every gather redirected to the own record
run() { SPH_LIBRARY=$PWD/pysph_amd/libsphhip_prof.so python bench.py --no-cpu-baseline --no-extras --no-counters --no-check --steps 20 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernel_ms_per_step']['pair'],4))"; }
for w in "" "--workload taylor_green" "--workload dam_break --dx 0.0055"; do
for a in 0 6 7 2 3; do echo "$w ablate $a: $(run $w --ablate $a)"; done
done
