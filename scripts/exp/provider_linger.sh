#!/bin/bash
# Does the exit of the data providers (their HIP runtimes going away while parties 1 and 2 stream the tables) cost table-phase
# time?  LINREG_PROVIDER_LINGER_MS=400 keeps the providers alive past the table phase; config 3 (TI rings) and config 2, 4 reps each.
for ms in 0 400 0 400; do
  echo "== LINREG_PROVIDER_LINGER_MS=$ms"
  LINREG_PROVIDER_LINGER_MS=$ms python3 scripts/startup_probe.py --configs c3-ti,c2 --reps 4 --out /tmp/linger_$ms.json 2>&1 | python3 -c "
import sys,re
cur=None; t={}
for l in sys.stdin:
    m=re.match(r'(\S+)\s+rep (\d+)\s+wall ([0-9.]+)', l)
    if m:
        if cur and 'a' in t and 'b' in t: print(cur, 'tables %.4f' % (t['b']-t['a']))
        cur='%s rep %s wall %s' % m.groups(); t={}
    m=re.match(r'\s+input_labels_in\s+([0-9.]+)', l)
    if m: t['a']=float(m.group(1))
    m=re.match(r'\s+tables_evaluated\s+([0-9.]+)', l)
    if m: t['b']=float(m.group(1))
if cur and 'a' in t and 'b' in t: print(cur, 'tables %.4f' % (t['b']-t['a']))
"
done
