#!/bin/bash
# shorter records in the matrix-vector launches (more rounds of the chip per launch: the small launches of the other chain get
# CUs at every round boundary), with and without high-priority queues for the small launches
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
run() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['exact_vs_oracle'], o['gate_steps_per_solve'])"
}
for w in 12288 24576 49152 98304; do
  for pr in 0 1; do
    export LGC_X_MV_WAVES=$w LGC_PRIO=$pr
    echo "== LGC_X_MV_WAVES=$w LGC_PRIO=$pr"; run
  done
done
