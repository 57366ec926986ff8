for v in A nowide; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"; python tests/tools/gpu_c5_sweep.py --check 1 2>&1 | tail -1 | cut -c1-200; python scripts/gpu_probe.py big mid 2>&1 | grep -E "^d=" | grep -v "0.35\|0.30"
done
