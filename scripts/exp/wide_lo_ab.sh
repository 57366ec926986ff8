#!/bin/bash
# waves per workgroup of the wide (one wave per record) launches: LGC_WIDE_LO = fewest waves gc_wide_waves may choose (12 = off)
#   the 64-lambda sweep (serialised G / E per op and the overlapped run), the headline solve, d = 100 and d = 250 Cholesky
for lo in 12 9 12 9 6 8; do
  echo "== LGC_WIDE_LO=$lo"
  LGC_WIDE_LO=$lo python scripts/exp/sweep_launch_profile.py 2>&1 | grep -E "^sweep|^DIV|^IPMAC|^MULSUB|^SUM|^HDIFF"
done
for lo in 12 9 12 9; do
  echo "== LGC_WIDE_LO=$lo (probe)"
  LGC_WIDE_LO=$lo python scripts/gpu_probe.py big chol250 2>&1 | grep -E "^d=" | cut -c1-110
done
