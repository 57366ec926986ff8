#!/bin/bash
# records per merged matrix-vector launch of a lambda sweep (d = 100, 64-bit): blocks of 8 and 64 circuits
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
for w in 12288 131072 524288 2097152; do
  export LGC_X_MV_WAVES=$w
  echo "== LGC_X_MV_WAVES=$w"; python scripts/dbg/block_probe.py 2>&1 | grep block
done
