#!/bin/bash
# timing experiment: which part of a gate level costs what (variant libraries with pieces switched off; results are wrong by design)
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_launch_profile.py 20 cholesky 0 2>&1 | grep -E "^d=|DIV|SQRT|MAC"
done
