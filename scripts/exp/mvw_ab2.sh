#!/bin/bash
export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_mvw.so
run() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['exact_vs_oracle'], o['gate_steps_per_solve'], o['launches_per_solve'], o['seconds_mac_garble_per_solve'])"
}
for w in ${WLIST:-65536 98304 131072 196608 262144}; do
    export LGC_X_MV_WAVES=$w LGC_PRIO=0
    echo "== LGC_X_MV_WAVES=$w"; run
done
