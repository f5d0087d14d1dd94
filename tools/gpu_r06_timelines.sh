#!/bin/bash
# kernel timelines of one steady-state step (rocprofv3 --kernel-trace) -> gpurun_out/r06t/timelines.txt
# (copied to profiles/r06_step_timelines.txt)
R=$(pwd); mkdir -p gpurun_out/r06t; O=$R/gpurun_out/r06t/timelines.txt
(echo "== 4 M cube (159^3), one steady-state step: start, gap to the previous kernel, duration [us] =="; bash tools/debug/trace_cube.sh
 echo; echo "== 1 M cube (100^3) =="; bash tools/debug/trace_cube.sh --n1 100
 echo; echo "== dam break dx 0.0055 (4.65 M particles, three arrays, merged order) =="; bash tools/debug/trace_cube.sh --workload dam_break --dx 0.0055
 echo; echo "== Taylor-Green 159^3 (periodic images without a round trip, real-particle wave tiles in the force pass) =="; bash tools/debug/trace_cube.sh --workload taylor_green) > $O 2>&1
cd /tmp && export TMPDIR=/tmp
for T in torch sphcomm; do
  rm -rf /tmp/prof
  SPH_HALO_TRANSPORT=$T rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o runc -- python $R/tools/debug/selfslab_rank.py > /tmp/log_$T.txt 2>&1
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  (echo; echo "== rank 4 of 8 of the 17.3 M dam break as its own periodic neighbour, transfers through $T (sphcomm: ncclSend/Recv on the context stream, the default of bench.py --gpus N) =="; python $R/tools/debug/trace_step.py $f; grep "^True" /tmp/log_$T.txt | cut -c1-120) >> $O
done
tail -40 $O
