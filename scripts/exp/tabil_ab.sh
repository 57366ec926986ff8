#!/bin/bash
# table rows of neighbouring records interleaved step by step (GC_X_TABIL): do the MAC kernels gain from writing / reading
# the tables of a workgroup's waves side by side?
run() {
  python bench.py --steps 3 --warmup 1 --no-traffic --no-e2e --no-sweep >/dev/null 2>&1; python -c "import sys,json; o=json.load(open('bench_detail.json')); print('bench', o['value'], o['ms_per_step'], o['exact_vs_oracle'], o['seconds_exclusive_per_solve'])"
}
for v in "" tabil16 tabil48 "" tabil16; do
  if [ -z "$v" ]; then unset LGC_LIB; else export LGC_LIB=$GRAFT_REPO_ROOT/scripts/exp/libs/lib_$v.so; fi
  echo "== ${v:-product}"; run
done
