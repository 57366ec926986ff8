#!/bin/bash
# GC_MAC_ADAPT=1 with at least 12 waves per workgroup (lib_macadapt12.so: -DGC_MAC_ADAPT=1 -DGC_MAC_ADAPT_LO_G=12 -DGC_MAC_ADAPT_LO_E=12)
# against the default build, over the shapes whose MAC launches are a few rounds of the chip
P="python scripts/exp/shape_probe.py"
L=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_${1:-macadapt12}.so
for cfg in "100 cgd 15 64" "64 cgd 15 64" "120 cgd 15 64" "40 cgd 15 64" "100 cholesky 0 64" "180 cholesky 0 64" "200 ldlt 0 64" "250 cholesky 0 32" "300 cgd 5 32" "100 cgd 15 32"; do
  for rep in 1 2; do
    $P $cfg 3 | cut -c1-20,40-70
    LGC_LIB=$L $P $cfg 3 | cut -c1-20,40-70 | sed 's/^/   adapt: /'
  done
done
