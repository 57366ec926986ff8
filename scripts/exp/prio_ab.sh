#!/bin/bash
# do high-priority queues for the chains' small launches (LGC_PRIO=1), alone and with a table ring that holds two
# needs a library built from the tree with scripts/exp/prio_streams.patch applied (git apply; LGC_PRIO is not in the product)
# matrix-vector products, shorten the d = 500 solve?  (The small launches of one chain run beside the other chain's MAC kernel.)
run() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['exact_vs_oracle'])"
}
for rep in 1 2; do
  unset LGC_PRIO LGC_RING_SLACK_MB
  echo "== default"; run
  export LGC_PRIO=1
  echo "== LGC_PRIO=1"; run
  export LGC_RING_SLACK_MB=61440
  echo "== LGC_PRIO=1 LGC_RING_SLACK_MB=61440"; run
  unset LGC_PRIO
  echo "== LGC_RING_SLACK_MB=61440"; run
done
