#!/bin/bash
# config 4 end to end by the number of parser threads per provider (LINREG_PARSE_THREADS); what the box gives a process
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
for t in 1 2 4 8 16 4 8; do
  export LINREG_PARSE_THREADS=$t
  python scripts/startup_probe.py --configs c4 --reps 2 --out gpurun_out/st_parse_$t.json > /dev/null 2>&1
  python - <<PY
import json
o=json.load(open("gpurun_out/st_parse_$t.json"))
for r in o:
    st=r['timeline']['steps']
    parsed=[t for p,v in r['timeline']['marks'].items() for t,m in v if 'parsed' in m]
    print("threads $t wall %.3f  hip_up %.3f parsed %.3f..%.3f phase1_done %.3f evaluated %.3f" % (r['phase12_wall_s'], st['hip_runtime_up']['t_s'], min(parsed), max(parsed), st['phase1_done']['t_s'], st['tables_evaluated']['t_s']))
PY
done
