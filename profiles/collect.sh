#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo
# root through gpurun):   bash profiles/collect.sh r01
# Pass 1: kernel trace + stats.  Passes 2..n: one PMC counter set per run
# (--pmc with --kernel-trace only, as the pool requires).  Raw output goes to
# gpurun_out/<tag>/; profiles/summarize.py condenses it into profiles/.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o runc -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_REQ_sum"; do
    NAME=$(echo $SET | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -o runc -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1
    echo "$NAME rc=$?" >> "$OUT/passes.log"
done
find "$OUT" -name "*_agent_info.csv" -delete
# counter CSVs of every kernel of every launch are large; keep the pair/pack/nnps kernels only
for f in $(find "$OUT" -name "*_counter_collection.csv"); do
    (head -1 "$f"; grep -E "k_pair|k_pack|k_nosrc|k_cell_keys|k_cell_start" "$f") > "$f.tmp" && mv "$f.tmp" "$f"
done
find "$OUT" -name "*_kernel_trace.csv" -path "*pmc_*" -delete
du -sh "$OUT"
