#!/bin/bash
# table cap of a merged sweep launch: 2^24 steps (product) against 2^25 (half as many launches per matrix-vector product)
for v in "" scap25 "" scap25; do
  if [ -z "$v" ]; then unset LGC_LIB; else export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so; fi
  echo "== ${v:-product}"; python scripts/dbg/block_probe.py 2>&1 | grep block
done
