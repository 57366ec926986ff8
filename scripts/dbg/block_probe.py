"""per-rank blocks of the 64-lambda sweep on one GPU: what an N-GPU run spends per rank (N = 64 / block)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import linreg_gc as lgc
import sweep
d, it = 100, 15
rng = np.random.default_rng(5)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
lams = sweep.c5_lambdas(64)
sysm = lgc.make_system(d, 64, 56, "cgd", it, 0.0, 2, 1, 0, 0)
for block in (8, 8, 64):
    t0 = time.perf_counter()
    s = lgc.Solver(sysm, seed=bytes(range(16)), lambdas=lams[:block])
    t1 = time.perf_counter()
    s.set_shares(shares); s.run(); b = s.beta()
    t2 = time.perf_counter()
    st = s.stats()
    s.run(profile=True); sx = s.stats()
    print("block %2d: create %.3f s run %.3f s (device %.3f); gates %.3e -> %.3e AND/s; exclusive garble %.3f eval %.3f (mac %.3f / %.3f)" % (
        block, t1 - t0, t2 - t1, st["seconds_total"], st["and_gates"], st["and_gates"] / (t2 - t1), sx["seconds_garble"], sx["seconds_eval"],
        sx["seconds_mac_garble"], sx["seconds_mac_eval"]), flush=True)
    s.close()
