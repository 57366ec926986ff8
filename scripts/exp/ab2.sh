#!/bin/bash
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_probe.py mid 2>&1 | grep -E "^d="
done
