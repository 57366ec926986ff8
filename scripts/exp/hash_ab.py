"""A/B timing of the engine over gate hash 1 (Chaskey-12): d=500 CGD-15 and the latency-bound shapes; LGC_LIB selects a
variant library (scripts/exp/build_variant.sh)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
kind = sys.argv[1] if len(sys.argv) > 1 else "chaskey12"
lgc.set_gate_hash(kind)
def run(d, alg, iters, reps=2):
    rng = np.random.default_rng(0)
    T = d * (d + 1) // 2
    shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
    s = lgc.Solver(lgc.make_system(d, 64, 56, alg, iters, 0.0, 2, 0, 0, 0)); s.set_shares(shares)
    best = None
    for _ in range(reps):
        s.run(); st = s.stats()
        if best is None or st["seconds_total"] < best["seconds_total"]: best = st
    print("%s d=%d %s-%d: %.4f s, %.3e AND/s; mac G %.3f E %.3f" % (kind, d, alg, iters, best["seconds_total"], best["and_gates"] / best["seconds_total"],
          best["seconds_mac_garble"], best["seconds_mac_eval"]), flush=True)
    s.close()
run(500, "cgd", 15, 3)
run(100, "cgd", 15, 3)
run(20, "cholesky", 0, 3)
