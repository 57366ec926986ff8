#!/bin/bash
# start-up cost of the HIP runtime under a few environment settings; solo and four processes at once
P=scripts/exp/bin/hip_init_probe
run() {  # label, env...
  local label="$1"; shift
  echo "== $label (solo)"; env "$@" $P | tr '\n' ';'; echo
  echo "== $label (4 at once, first process shown)"
  for i in 1 2 3; do env "$@" $P > /dev/null & done
  env "$@" $P | tr '\n' ';'; echo; wait
}
run baseline A=1
run baseline-again A=1
run no-sdma HSA_ENABLE_SDMA=0
run hwq2 GPU_MAX_HW_QUEUES=2
run visible HIP_VISIBLE_DEVICES=0 ROCR_VISIBLE_DEVICES=0
run nointr HSA_ENABLE_INTERRUPT=0
run nofrag HSA_DISABLE_FRAGMENT_ALLOCATOR=1
echo "== big allocations (fresh, then again right after the free)"
$P 4e9 | tail -3 | tr '\n' ';'; echo
$P 4e9 | tail -3 | tr '\n' ';'; echo
$P 12e9 | tail -3 | tr '\n' ';'; echo
$P 12e9 | tail -3 | tr '\n' ';'; echo
sleep 2
$P 12e9 | tail -3 | tr '\n' ';'; echo
