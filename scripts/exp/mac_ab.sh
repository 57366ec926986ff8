#!/bin/bash
for v in A Q; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"; python scripts/gpu_launch_profile.py 1 cgd 1 | grep -E "^DIV|^MAX|^IPMAC|^MULSUB|total"; python scripts/gpu_launch_profile.py 100 cgd 15 | grep -E "^DIV|total"; python scripts/gpu_launch_profile.py 20 cholesky 0 | grep -E "^DIV|^SQRT|total"
done
