#!/bin/bash
# records per launch of the big multiply-accumulate launches: CGD (LGC_X_MV_WAVES) and the factorisations (LGC_X_CHOL_WAVES)
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
P="python scripts/exp/shape_probe.py"
for w in 12288 32768 131072; do export LGC_X_MV_WAVES=$w; $P 200 cgd 15 64; done
for w in 12288 32768 131072; do export LGC_X_MV_WAVES=$w; $P 300 cgd 15 64; done
for w in 12288 49152 131072 262144; do export LGC_X_MV_WAVES=$w; $P 500 cgd 20 32; done
unset LGC_X_MV_WAVES
for w in 4096 8192 16384 32768; do export LGC_X_CHOL_WAVES=$w; $P 250 cholesky 0 64; done
for w in 4096 16384 65536; do export LGC_X_CHOL_WAVES=$w; $P 500 cholesky 0 64 1; done
