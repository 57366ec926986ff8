#!/bin/bash
# hash_ab.sh VARIANT... : scripts/exp/hash_ab.py under each variant library (A = the product library)
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/exp/hash_ab.py chaskey12 2>&1 | grep -v "^devices"
done
