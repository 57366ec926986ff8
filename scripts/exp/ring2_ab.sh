#!/bin/bash
# does a table ring that holds TWO matrix-vector products (garble k + 1 beside evaluate k) help the d = 500 solve?
for slack in "" 61440 "" 61440; do
  if [ -z "$slack" ]; then unset LGC_RING_SLACK_MB; else export LGC_RING_SLACK_MB=$slack; fi
  echo "== LGC_RING_SLACK_MB=${slack:-default}"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['exact_vs_oracle'])"
done
