#!/bin/bash
# Karatsuba records in the matrix-vector products of smaller systems (now that a record may hold a single pair): d = 100 and 64
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
P="python scripts/exp/shape_probe.py"
for cfg in "100 cgd 15 64" "64 cgd 15 64" "40 cgd 15 64"; do
  for k in "" 4096 1024 256; do
    if [ -z "$k" ]; then unset LGC_X_KARA_MIN; else export LGC_X_KARA_MIN=$k; fi
    $P $cfg 4
  done
done
