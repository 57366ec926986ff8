#!/bin/bash
# ring_modes_ab.sh with the one-stream asynchronous mode (LINREG_RING_ASYNC=2) beside sync (0) and two-stream async (1):
# the table phase on the evaluator's marks AND the wall clock of the slower process (queue creation and exit included)
#   bash scripts/exp/ring_modes3_ab.sh <d> <alg> <iters> [repeats]
R=${GRAFT_REPO_ROOT:-$PWD}
EXE=$R/linreg-mpc_amd/host/bin/test_linear_system
D=$1; ALG=$2; IT=$3; REP=${4:-5}
python3 $R/scripts/exp/two_proc_input.py $D /tmp/ls_$D.in
for rep in $(seq $REP); do
  for mode in 0 1 2; do
    export LINREG_RING_ASYNC=$mode
    PORT=$((20000 + RANDOM % 5000))
    T0=$(date +%s.%N)
    $EXE $PORT 1 /tmp/ls_$D.in $ALG $IT 56 --host=127.0.0.1 --table_ring > /tmp/p1.out 2>&1 &
    LINREG_TRACE=1 $EXE $PORT 2 /tmp/ls_$D.in $ALG $IT 56 --host=127.0.0.1 --table_ring > /tmp/p2.out 2> /tmp/p2.err
    wait
    T1=$(date +%s.%N)
    python3 - $mode $T0 $T1 <<'PY'
import re, sys
t = {}
for m in re.finditer(r"^LGCT \S+ ([0-9.]+) (.*)$", open("/tmp/p2.err").read(), re.M):
    t.setdefault(m.group(2), float(m.group(1)))
a, b = t.get("first table evaluated"), t.get("tables evaluated")
print("mode%s tables %.4f wall %.3f" % (sys.argv[1], (b - a) if a and b else -1, float(sys.argv[3]) - float(sys.argv[2])))
PY
  done
done | sort | awk '{k=$1; v[k]=v[k]" "$3; w[k]=w[k]" "$5} END {for (k in v) print k, "tables", v[k], " wall", w[k]}'
