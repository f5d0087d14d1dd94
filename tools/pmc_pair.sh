#!/bin/bash
# PMC passes over the pair kernel of one bench configuration (development aid).
#   tools/pmc_pair.sh <tag> <bench args...>      -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/p$i" -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/p$i.log" 2>&1
    echo "pass $i rc=$?" >> "$OUT/passes.log"
done
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_pair' not in k: continue
        acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in acc.items():
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            # one row per dispatch (and possibly per dimension): mean per launch
            fo.write('  %-40s %.6g  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
P
rm -rf "$OUT"/p*/
