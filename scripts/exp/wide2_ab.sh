#!/bin/bash
# the 500-record launches of the chains (dividers, updates) in the one-wave-per-record kernel instead of the 4-wave kernel:
# slower alone, but they run beside the other chain's MAC kernel and take a fifth of the CUs
run() {
  python bench.py --steps 3 --warmup 1 --no-traffic --no-e2e --no-sweep --no-cpu-baseline >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['seconds_exclusive_per_solve'])"
}
for v in "" wide450 wide300 "" wide450 wide300; do
  if [ -z "$v" ]; then unset LGC_LIB; else export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so; fi
  echo "== ${v:-product}"; run
done
