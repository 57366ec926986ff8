#!/bin/bash
# OT-mode phase 1 of config 3 (1.6e9 extended OTs between two provider PROCESSES on one GPU): batch size
for lg in 25 24 23 22 25 24 23; do
  echo "== LINREG_OT_BATCH_LOG2=$lg"
  LINREG_OT_BATCH_LOG2=$lg python scripts/startup_probe.py --configs c3-ot --reps 2 2>&1 | grep -E "wall|phase1_done"
done
