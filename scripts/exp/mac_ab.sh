#!/bin/bash
for v in A S0 E1024 E512 G768; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"; python scripts/gpu_probe.py big | tail -3
done
