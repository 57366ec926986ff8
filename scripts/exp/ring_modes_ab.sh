#!/bin/bash
# The table phase of a two-process solve (bin/test_linear_system --table_ring), evaluator's trace marks "first table
# evaluated" -> "tables evaluated", for the ways the CSP can drive its launches: sync (pass behind its record kernel, a
# device synchronisation and a token per launch), async (two queues + notifier thread: lgc_party_garble_ring_begin / _wait),
# and -- only with scripts/exp/lazy_pass.patch applied -- lazy (one queue, the table pass owed to the next launch's grid:
# measured in round 5, no gain, not in the tree).  Six runs each on one box: d = 100 CGD-15 sync 0.1465-0.1482 s, async
# 0.1397-0.1471, lazy 0.1496-0.1520; d = 20 Cholesky 0.0447-0.0449 / 0.0435-0.0438 / 0.0443-0.0445; d = 200 Cholesky
# 1.34-1.38 / 1.29-1.31 / 1.34-1.37.
#   bash scripts/exp/ring_modes_ab.sh <d> <alg> <iters> [repeats]
R=${GRAFT_REPO_ROOT:-$PWD}
EXE=$R/linreg-mpc_amd/host/bin/test_linear_system
D=$1; ALG=$2; IT=$3; REP=${4:-5}
python3 $R/scripts/exp/two_proc_input.py $D /tmp/ls_$D.in
for rep in $(seq $REP); do
  for mode in ${MODES:-sync async}; do
    case $mode in lazy) export LINREG_RING_ASYNC=0 LINREG_RING_LAZY=1;; sync) export LINREG_RING_ASYNC=0 LINREG_RING_LAZY=0;; async) export LINREG_RING_ASYNC=1;; esac
    PORT=$((20000 + RANDOM % 5000))
    $EXE $PORT 1 /tmp/ls_$D.in $ALG $IT 56 --host=127.0.0.1 --table_ring > /tmp/p1.out 2>&1 &
    LINREG_TRACE=1 $EXE $PORT 2 /tmp/ls_$D.in $ALG $IT 56 --host=127.0.0.1 --table_ring > /tmp/p2.out 2> /tmp/p2.err
    wait
    python3 - $mode <<'PY'
import re, sys
t = {}
for m in re.finditer(r"^LGCT \S+ ([0-9.]+) (.*)$", open("/tmp/p2.err").read(), re.M):
    t.setdefault(m.group(2), float(m.group(1)))
a, b = t.get("first table evaluated"), t.get("tables evaluated")
print("%-5s tables %.4f s" % (sys.argv[1], b - a) if a and b else "%-5s (marks missing)" % sys.argv[1])
PY
  done
done | sort | awk '{k=$1; v[k]=v[k]" "$3; n[k]++} END {for (k in v) print k, v[k]}'
