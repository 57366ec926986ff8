#!/bin/bash
# Karatsuba MAC launches of a few rounds with fewer waves per workgroup (GC_MACK_ADAPT build) against the fixed 16 / 12:
# the factorisations at d = 250 and d = 500 (one launch per column, 1-5 rounds each), and the d = 500 CGD headline (unaffected
# by construction: its launches are 30 rounds)
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_probe.py chol250 2>&1 | grep "^d=250\|profiled"
  python tests/tools/gpu_big_cholesky.py 2>&1 | grep "w=64"
done
