#!/bin/bash
for v in "$@"; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"
  python scripts/gpu_probe.py big 2>&1 | grep -E "^d=" | head -1
  python tests/tools/gpu_c5_sweep.py --lambdas 8 --check 2 2>&1 | tail -1 | cut -c1-220
  python tests/tools/gpu_c5_sweep.py --lambdas 64 --check 1 2>&1 | tail -1 | cut -c1-220
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'])"
done
