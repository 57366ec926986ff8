#!/bin/bash
for v in J K L M; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"; python scripts/gpu_launch_profile.py 500 cgd 1 | grep -E "^MAC|total"
done
