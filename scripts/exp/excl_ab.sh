#!/bin/bash
# GC_MAC_EXCLUSIVE (a big MAC launch waits for the evaluation of the previous big MAC launch) with the short records of round 4
for v in "" noexcl "" noexcl; do
  if [ -z "$v" ]; then unset LGC_LIB; else export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so; fi
  echo "== ${v:-product}"
  python bench.py --steps 2 --warmup 1 --no-traffic --no-e2e --no-cpu-baseline >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); s=o['sweep64']; print('bench', o['ms_per_step'], {k: s[k] for k in s if 'second' in k or k=='exact_vs_oracle'})"
done
