#!/bin/bash
# exclusive (serialised) kernel times of the matrix-vector launches by record length
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
for c in 2 4 8 22; do
  export LGC_X_MV_WAVES=131072 LGC_X_MV_CHUNK=$c
  echo "== chunk $c"; python scripts/gpu_launch_profile.py 500 cgd 3 | grep -E "total|MACK|SUM"
done
