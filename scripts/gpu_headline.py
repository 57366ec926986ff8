"""the headline solve (d = 500 CGD-15, 64-bit, both roles on one GPU) a few times: seconds per solve (scripts/exp A/B runs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
d = int(sys.argv[1]) if len(sys.argv) > 1 else 500
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
alg = sys.argv[4] if len(sys.argv) > 4 else "cgd"
w = int(sys.argv[5]) if len(sys.argv) > 5 else 64
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
if w == 32:
    shares &= np.uint64(0xffffffff)
s = lgc.Solver(lgc.make_system(d, w, 56 if w == 64 else 30, alg, iters if alg == "cgd" else 0, 0.0, 2, 0, 0, 0))
s.set_shares(shares)
s.run()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t0)
st = s.stats()
print("d=%d %s-%d w=%d: %s s per solve (min %.4f); MAC garble %.4f s per solve; %s" % (d, alg, iters, w, " ".join("%.4f" % t for t in ts), min(ts),
      st["seconds_mac_garble"], " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("LGC_"))))
s.close()
