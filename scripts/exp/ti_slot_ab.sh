for mb in 64 256 512 1024; do
  echo "== LINREG_TI_SLOT_MB=$mb"
  LINREG_TI_SLOT_MB=$mb python scripts/startup_probe.py --configs c4 --reps 2 2>&1 | grep -E "wall|phase1_done|barrier  "
done
