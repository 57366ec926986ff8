"""solve time of one system for the launch-shape experiments: shape_probe.py d alg iters w [reps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
d, alg, iters, w = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
p = 56 if w == 64 else 30
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
s = lgc.Solver(sysm)
s.set_shares(shares)
ts = []
for _ in range(reps):
    s.run()
    st = s.stats()
    ts.append(st["seconds_total"])
print("d=%d %s-%d w=%d env %s: %s s; steps %d launches %d; mac G %.3f" % (d, alg, iters, w,
      {k: v for k, v in os.environ.items() if k.startswith("LGC_X") or k in ("LGC_PRIO", "LGC_RING_SLACK_MB")},
      " ".join("%.4f" % t for t in ts), st["gate_steps"], st["launches"], st["seconds_mac_garble"]), flush=True)
s.close()
