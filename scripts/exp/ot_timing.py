import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bench
os.environ["LINREG_TIMING"] = "1"
bench.phase12_wall(np, "warm-up", 200, 4, [0, 2], "cholesky", 0, ["--table_ring"], 0)
import subprocess
# rerun keeping stderr: patch startup_timeline to dump raw stderr
orig = bench.startup_timeline
def dump(stderrs, m0):
    for k, t in enumerate(stderrs):
        for l in t.splitlines():
            if not l.startswith("LGCT") or "phase" in l or "data on" in l: print("p%d: %s" % (k + 1, l))
    return orig(stderrs, m0)
bench.startup_timeline = dump
r = bench.phase12_wall(np, "c3-ot", 10000, 100, [0, 50], "cgd", 15, ["--ot_ring", "--input_ring", "--table_ring"], 0)
print(r["phase12_wall_s"])
