"""diagnostic: where the time of an 8-lambda merged solve goes, repeated (creation / input / run / close)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import torch  # noqa: F401  (HIP runtime load order)
import numpy as np
import linreg_gc as lgc
import sweep
d, iters, w, p, P = 100, 15, 64, 56, 2
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lams = sweep.c5_lambdas(64)[:nl]
T = d * (d + 1) // 2
rng = np.random.default_rng(1)
sh = rng.integers(0, 2**62, size=(P, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, w, p, "cgd", iters, 0.0, P, 1, 0, 0)
for rep in range(6):
    t0 = time.time(); sv = lgc.Solver(sysm, seed=os.urandom(16), lambdas=lams)
    t1 = time.time(); sv.set_shares(sh)
    t2 = time.time(); sv.run()
    t3 = time.time(); st = sv.stats(); sv.close()
    t4 = time.time()
    print("rep %d: create %.3f set_shares %.3f run %.3f (device %.3f) close %.3f" % (rep, t1 - t0, t2 - t1, t3 - t2, st["seconds_total"], t4 - t3), flush=True)
