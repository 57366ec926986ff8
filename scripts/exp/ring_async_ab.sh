#!/bin/bash
# two-process solves (bin/test_linear_system --table_ring) with the garbler's launches enqueued asynchronously (default since
# round 5) against the loop of rounds 2-4 (LINREG_RING_ASYNC=0: garble, device synchronisation, token, next launch)
R=${GRAFT_REPO_ROOT:-$PWD}
EXE=$R/linreg-mpc_amd/host/bin/test_linear_system
run() {  # d alg iters
  python3 $R/scripts/exp/two_proc_input.py $1 /tmp/ls_$1.in
  for mode in async sync async sync; do
    [ $mode = sync ] && export LINREG_RING_ASYNC=0 || export LINREG_RING_ASYNC=1
    PORT=$((20000 + RANDOM % 5000))
    $EXE $PORT 1 /tmp/ls_$1.in $2 $3 56 --host=127.0.0.1 --table_ring > /tmp/p1.out 2>&1 &
    $EXE $PORT 2 /tmp/ls_$1.in $2 $3 56 --host=127.0.0.1 --table_ring > /tmp/p2.out 2>&1
    wait
    echo "d=$1 $2-$3 $mode: $(grep -E "Iteration $(($3-1)) time|Time elapsed|OT time" /tmp/p2.out | tr '\n' ' ')"
  done
}
run 100 cgd 15
run 20 cholesky 0
run 200 cholesky 0
