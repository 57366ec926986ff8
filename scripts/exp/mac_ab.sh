#!/bin/bash
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  [ -n "$AB_PROBE" ] && python scripts/gpu_probe.py mid 2>&1 | grep "^d=100" | head -3
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['seconds_mac_garble_per_solve'], o['seconds_exclusive_per_solve'])"
done
