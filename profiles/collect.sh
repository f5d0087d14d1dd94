#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo
# root through gpurun):   bash profiles/collect.sh r02 [bench args...]
#   gpurun_out/<tag>/<workload>/stats/...  kernel trace + stats of `python bench.py <args>`
#   gpurun_out/<tag>/<workload>/pmc_<set>  one PMC counter set per run (--pmc with
#                                          --kernel-trace only, as the pool requires)
# profiles/summarize.py condenses the raw CSVs into profiles/.
set -u
TAG=${1:-r05}; shift || true
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
# counter sets: 1 = all of them, 2 = the traffic / busy sets only
run_one() {   # name, pmc (0/1/2), bench args...
    local NAME=$1 PMC=$2; shift 2
    local OUT=$ROOT/gpurun_out/$TAG/$NAME
    mkdir -p "$OUT"
    local BENCH="python $ROOT/bench.py --no-cpu-baseline --no-check --no-extras --no-counters $*"
    $BENCH --steps 32 --warmup 6 > "$OUT/bench.json" 2> "$OUT/bench.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o runc -- $BENCH --steps 32 --warmup 6 > "$OUT/stats.log" 2>&1
    if [ "$PMC" = 2 ]; then
        for SET in "FETCH_SIZE" "WRITE_SIZE" \
                   "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
                   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
                   "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
            local N=$(echo $SET | cut -d' ' -f1)
            timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o runc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc_$N.log" 2>&1
            echo "$N rc=$?" >> "$OUT/passes.log"
        done
    fi
    if [ "$PMC" = 1 ]; then
        for SET in "FETCH_SIZE" "WRITE_SIZE" \
                   "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
                   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
                   "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
                   "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
                   "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
                   "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_REQ_sum"; do
            local N=$(echo $SET | cut -d' ' -f1)
            timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o runc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc_$N.log" 2>&1
            echo "$N rc=$?" >> "$OUT/passes.log"
        done
    fi
    find "$OUT" -name "*_agent_info.csv" -delete
    # counter CSVs of every kernel of every launch are large; keep the hot kernels only
    for f in $(find "$OUT" -name "*_counter_collection.csv"); do
        (head -1 "$f"; grep -E "k_pair|k_pack|k_nosrc|k_bin_keys|k_bucket" "$f") > "$f.tmp" && mv "$f.tmp" "$f"
    done
    find "$OUT" -name "*_kernel_trace.csv" -path "*pmc_*" -delete
}
if [ $# -gt 0 ]; then
    run_one custom 1 "$@"
else
    run_one cube 1
    run_one cube_f32 2 --dtype f32
    run_one taylor_green 2 --workload taylor_green
    run_one rings 2 --workload elastic
    run_one rings_f32 0 --workload elastic --dtype f32
    run_one dam_break 0 --workload dam_break
    run_one dam_break_4m 2 --workload dam_break --dx 0.0055
    run_one dam_break_4m_vh 0 --workload dam_break --dx 0.0055 --vary-h 0.15
    run_one cube_vh 0 --vary-h 0.15
    # (the 16 M dam break is timed and checked by the default bench line: extra.secondary)
fi
du -sh "$ROOT/gpurun_out/$TAG"
