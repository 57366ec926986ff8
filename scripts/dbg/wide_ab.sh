#!/bin/bash
# A/B: from how many records a generic launch runs one wave per record (GC_WIDE_LAUNCH)
for v in A wide520 wide1024; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/dbg/block_probe.py 2>&1 | grep "block  8" | tail -1
  python scripts/gpu_launch_profile.py 500 cgd 2 2>&1 | grep -E "^d=|DIV|MAX|ABS|MULSUB|IPM"
done
