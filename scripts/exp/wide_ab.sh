for v in A soloinl; do
  if [ "$v" = "A" ]; then unset LGC_LIB; else export LGC_LIB=$PWD/scripts/exp/libs/lib_$v.so; fi
  echo "== variant $v"; python tests/tools/gpu_c5_sweep.py --check 1 2>&1 | tail -1 | cut -c1-200; python scripts/gpu_probe.py big 2>&1 | grep -E "^d=" | head -1
done
