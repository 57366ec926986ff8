#!/bin/bash
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
P="python scripts/exp/shape_probe.py"
for w in 24576 32768 49152 65536 98304; do export LGC_X_MV_WAVES=$w; $P 500 cgd 20 32 3; done
export LGC_X_MV_WAVES=131072
$P 500 cgd 15 64 3
LGC_RING_SLACK_MB=61440 $P 500 cgd 15 64 3
LGC_RING_SLACK_MB=61440 LGC_PRIO=1 $P 500 cgd 15 64 3
LGC_PRIO=1 $P 500 cgd 15 64 3
for w in 131072 262144; do export LGC_X_CHOL_WAVES=$w; $P 500 cholesky 0 64 1; done
export LGC_X_CHOL_WAVES=65536
$P 400 ldlt 0 64 1
unset LGC_X_CHOL_WAVES
$P 400 ldlt 0 64 1
