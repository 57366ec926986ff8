#!/bin/bash
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python tests/tools/gpu_c5_sweep.py --lambdas 8 --check 2 2>&1 | tail -1 | cut -c1-200
  python tests/tools/gpu_c5_sweep.py --lambdas 64 --check 1 2>&1 | tail -1 | cut -c1-200
  python scripts/gpu_launch_profile.py 500 cgd 1 2>&1 | grep -E "^d=|SUM|DIV"
done
