#!/bin/bash
# Karatsuba records in the matrix-vector product of small systems TOGETHER with waves per workgroup chosen by rounds x waves
# (lib_kadapt10 / lib_kadapt12: -DGC_X_MV_WAVES_ENV -DGC_MACK_ADAPT=1 -DGC_MACK_ADAPT_LO_G/E=10|12 -DGC_MACK_ADAPT_MAX_ROUNDS=3)
P="python scripts/exp/shape_probe.py"
for cfg in "100 cgd 15 64" "64 cgd 15 64" "120 cgd 15 64"; do
  for rep in 1 2; do
    $P $cfg 3 | cut -c1-20,40-70
    for v in kadapt10 kadapt12; do for k in 2048; do
      LGC_X_KARA_MIN=$k LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so $P $cfg 3 | cut -c1-20,60-130 | sed "s/^/   $v kara_min=$k: /"
    done; done
  done
done
for cfg in "250 cholesky 0 64" "500 cgd 2 64"; do
  $P $cfg 2 | cut -c1-22,40-70
  for v in kadapt10 kadapt12; do LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so $P $cfg 2 | cut -c1-22,40-70 | sed "s/^/   $v: /"; done
done
