#!/bin/bash
# mid-size systems: old record target (12 288) against the new defaults (131 072 at 64 bit, 65 536 at 32 bit)
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
P="python scripts/exp/shape_probe.py"
for cfg in "150 cgd 15 64" "250 cgd 15 64" "400 cgd 15 64" "150 cgd 20 32" "200 cgd 20 32" "300 cgd 20 32" "400 cgd 20 32"; do
  for w in 12288 ""; do
    if [ -z "$w" ]; then unset LGC_X_MV_WAVES; else export LGC_X_MV_WAVES=$w; fi
    $P $cfg 3
  done
done
