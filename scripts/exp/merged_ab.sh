#!/bin/bash
# column-split kernel: one wave per sixteen gates hashing both operands (GC_SPLIT_MERGED) against two waves
for v in A merged A merged; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_launch_profile.py 20 cholesky 0 2>&1 | grep -E "^d=|DIV|SQRT"
  python scripts/gpu_launch_profile.py 100 cgd 15 2>&1 | grep -E "^d=|DIV"
done
