#!/bin/bash
# per-launch hand-off latency of the two-process runs: interrupt-driven waits (default) against polling (HSA_ENABLE_INTERRUPT=0)
for v in default nointr default nointr; do
  if [ "$v" = nointr ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  echo "== $v"
  python scripts/startup_probe.py --configs c3-ti,c2 --reps 2 2>&1 | grep -E "wall|first_table|tables_evaluated"
done
