"""a merged lambda sweep on one GPU, timed (scripts/exp A/B runs): d, lambdas, iterations"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import torch  # noqa: F401  (HIP runtime load order)
import numpy as np
import linreg_gc as lgc
import sweep
d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 15
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lams = sweep.c5_lambdas(64)[:nl]
T = d * (d + 1) // 2
rng = np.random.default_rng(1)
sh = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, 64, 56, "cgd", iters, 0.0, 2, 1, 0, 0)
print("creating", flush=True)
sv = lgc.Solver(sysm, seed=bytes(range(16)), lambdas=lams)
print("created", flush=True)
sv.set_shares(sh)
ts = []
for rep in range(reps + 1):
    t0 = time.perf_counter(); sv.run(); ts.append(time.perf_counter() - t0)
    print("run %d: %.4f s" % (rep, ts[-1]), flush=True)
print("d=%d x %d lambdas CGD-%d: %s s per sweep (min %.4f); %s" % (d, nl, iters, " ".join("%.4f" % t for t in ts[1:]), min(ts[1:]),
      " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("LGC_"))))
sv.close()
