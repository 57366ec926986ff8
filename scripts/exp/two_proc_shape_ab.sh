#!/bin/bash
# two-process d = 500 CGD-15 (bin/test_linear_system --table_ring): launch size (--table_chunk_mb) x run-ahead room of the byte
# ring (LGC_PARTY_RING_SLACK_MB).  Args: chunk_mb:slack_mb ...   Prints the evaluator's last iteration clock.
R=${GRAFT_REPO_ROOT:-$PWD}; D=${D:-500}; IT=${IT:-15}
python3 $R/scripts/exp/two_proc_input.py $D /tmp/ls_$D.in
EXE=$R/linreg-mpc_amd/host/bin/test_linear_system
for v in "$@"; do
  c=${v%%:*}; s=${v##*:}
  for rep in 1 2; do
    PORT=$((20000 + RANDOM % 5000))
    opt=""; [ "$c" != "0" ] && opt="--table_chunk_mb=$c"
    env="" ; [ "$s" != "0" ] && export LGC_PARTY_RING_SLACK_MB=$s || unset LGC_PARTY_RING_SLACK_MB
    $EXE $PORT 1 /tmp/ls_$D.in cgd $IT 56 --host=127.0.0.1 --table_ring $opt > /tmp/p1.out 2>&1 &
    $EXE $PORT 2 /tmp/ls_$D.in cgd $IT 56 --host=127.0.0.1 --table_ring $opt > /tmp/p2.out 2>&1
    wait
    echo "chunk_mb=$c slack_mb=$s: $(grep -E "Iteration $((IT-1)) time|Time elapsed" /tmp/p2.out | tr '\n' ' ') $(tail -1 /tmp/p1.out | cut -c1-80)"
  done
done
