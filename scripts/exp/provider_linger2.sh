#!/bin/bash
# provider_linger.sh with the providers leaving about when the table phase ends (config 2: 70 ms, config 3: 180 ms): the wall clock
run() { LINREG_PROVIDER_LINGER_MS=$1 python3 scripts/startup_probe.py --configs $2 --reps 6 --out /tmp/l.json 2>&1 | python3 -c "
import sys,re
cur=None; t={}; rows=[]
for l in sys.stdin:
    m=re.match(r'(\S+)\s+rep (\d+)\s+wall ([0-9.]+)', l)
    if m:
        if cur and 'a' in t and 'b' in t: rows.append((cur, t['b']-t['a']))
        cur=float(m.group(3)); t={}
    m=re.match(r'\s+input_labels_in\s+([0-9.]+)', l)
    if m: t['a']=float(m.group(1))
    m=re.match(r'\s+tables_evaluated\s+([0-9.]+)', l)
    if m: t['b']=float(m.group(1))
if cur and 'a' in t and 'b' in t: rows.append((cur, t['b']-t['a']))
print('$2 linger $1 ms: wall', ' '.join('%.3f' % w for w, _ in sorted(rows)), ' tables', ' '.join('%.4f' % x for _, x in rows))
"; }
for rep in 1 2; do run 0 c2; run 70 c2; run 0 c3-ti; run 180 c3-ti; done
