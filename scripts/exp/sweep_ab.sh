#!/bin/bash
# A/B of library variants (scripts/exp/build_variant.sh) on merged lambda sweeps of config 5 and one d=500 iteration
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  for nl in 8 16 64; do python tests/tools/gpu_c5_sweep.py --lambdas $nl --check 1 2>&1 | tail -1 | cut -c1-140; done
  python scripts/gpu_launch_profile.py 500 cgd 1 2>&1 | grep -E "^d=|SUM|DIV"
done
