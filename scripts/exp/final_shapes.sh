#!/bin/bash
# the shapes touched by the end-of-round-5 launch rules (GC_MAC_ADAPT, GC_MACK_ADAPT <= 3 rounds, Karatsuba from d = 91), final build
P="python scripts/exp/shape_probe.py"
for cfg in "100 cgd 15 64" "95 cgd 15 64" "110 cgd 15 64" "120 cgd 15 64" "150 cgd 10 64" "200 cgd 5 64" "64 cgd 15 64" "100 cholesky 0 64" "190 cholesky 0 64" "250 cholesky 0 64" "200 ldlt 0 64"; do
  $P $cfg 3 | cut -c1-24,40-75
done
python tests/tools/gpu_big_cholesky.py 2>&1 | tail -4
