#!/bin/bash
# Kernel timeline of a TWO-PROCESS solve (CSP and Evaluator = bin/test_linear_system 1 / 2, --table_ring) beside the
# co-located one: each process under its own rocprofv3 --kernel-trace, merged on the GPU's clock by scripts/dbg/timeline.py.
#   bash scripts/exp/two_proc_trace.sh <d> <iters> <out dir under gpurun_out>
R=$GRAFT_REPO_ROOT; D=${1:-500}; IT=${2:-3}; O=$R/gpurun_out/${3:-twoproc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/exp/two_proc_input.py $D /tmp/ls_$D.in
PORT=$((20000 + RANDOM % 5000))
EXE=$R/linreg-mpc_amd/host/bin/test_linear_system
rocprofv3 --kernel-trace --output-format csv -d $O/p1 -- $EXE $PORT 1 /tmp/ls_$D.in cgd $IT 56 --host=127.0.0.1 --table_ring > $O/p1.out 2>&1 &
P1=$!
rocprofv3 --kernel-trace --output-format csv -d $O/p2 -- $EXE $PORT 2 /tmp/ls_$D.in cgd $IT 56 --host=127.0.0.1 --table_ring > $O/p2.out 2>&1
wait $P1
grep -E "Iteration|Time elapsed" $O/p2.out | tail -5
python3 $R/scripts/dbg/timeline.py $O/p1 $O/p2 400 > $O/timeline_two_process.txt 2>&1
# the co-located solve of the same size, same trace
rocprofv3 --kernel-trace --output-format csv -d $O/co -- python3 $R/scripts/gpu_probe.py trace$D > $O/co.out 2>&1
python3 $R/scripts/dbg/timeline.py $O/co 400 > $O/timeline_co_located.txt 2>&1
find $O -name "*.csv" -size +2M -delete
head -12 $O/timeline_two_process.txt; head -8 $O/timeline_co_located.txt
