#!/bin/bash
# where the HIP runtime's start-up goes: its own log with timestamps, solo and with three competitors
P=scripts/exp/bin/hip_init_probe
$P > /dev/null   # page in
sleep 1
echo "== solo, AMD_LOG_LEVEL=4 (first/last lines and the largest gaps)"
AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x7fffffff $P 2> /tmp/hiplog_solo.txt | head -3
wc -l /tmp/hiplog_solo.txt
python3 - <<'PY'
import re
rows=[]
for l in open('/tmp/hiplog_solo.txt', errors='replace'):
    m=re.search(r'\[\s*(\d+)\s*(?:us)?\]|ts:\s*(\d+)|:(\d{6,}) us', l)
    m2=re.search(r'(\d{9,}) us', l)
    if m2: rows.append((int(m2.group(1)), l.strip()[:160]))
rows.sort()
print("lines with timestamps:", len(rows))
if rows:
    t0=rows[0][0]
    gaps=sorted(((rows[i+1][0]-rows[i][0], i) for i in range(len(rows)-1)), reverse=True)[:12]
    for g,i in sorted(gaps, key=lambda x:x[1]):
        print("gap %8.1f ms after [%8.1f ms] %s" % (g/1e3, (rows[i][0]-t0)/1e3, rows[i][1][:120]))
        print("                         next: %s" % rows[i+1][1][:120])
PY
head -5 /tmp/hiplog_solo.txt
echo "== LD_DEBUG=statistics"
LD_DEBUG=statistics $P 2>&1 | grep -E "total startup|relocation|load" | head -8
