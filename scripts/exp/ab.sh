#!/bin/bash
# A/B of library variants (scripts/exp/libs/lib_<v>.so; A = the in-tree build): launch profiles + d=500 probe
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_launch_profile.py 20 cholesky 2>&1 | grep -E "^d=|^DIV|^SQRT|^MAC|^SUBSUM"
  python scripts/gpu_launch_profile.py 100 cgd 15 2>&1 | grep -E "^d=|^DIV|^MULSUB|^MAX|^IPM|^SUM"
  python scripts/gpu_probe.py big mid 2>&1 | grep -E "^d="
done
