"""one-off: d=500 Cholesky / LDL^T (64-bit) and d=500 Cholesky 32-bit vs the oracle, with timing
   python tests/tools/gpu_big_cholesky.py [aes128|chaskey12]   (the gate hash, lgc_set_gate_hash)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linreg_gc as lgc, orc
from helpers import oracle_solve, split_shares, synth_system
oracle = orc.load()
if len(sys.argv) > 1:
    lgc.set_gate_hash(sys.argv[1])
print("gate hash:", lgc.gate_hash(), flush=True)
for (d, w, p, alg) in ((500, 64, 56, "cholesky"), (500, 64, 56, "ldlt"), (500, 32, 30, "cholesky")):
    rng = np.random.default_rng(d + w)
    A, b = synth_system(oracle, rng, 3 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, 0, 1e-3, 2, 1, 0, 0)
    t0 = time.time(); s = lgc.Solver(sysm); s.set_shares(shares); t1 = time.time(); s.run(); t2 = time.time()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, 0, 1e-3, 1)
    st = s.stats()
    print("d=%d w=%d %s: build %.1fs run %.2fs gates %.3e (%.3e AND/s) launches %d exact=%s" % (
        d, w, alg, t1 - t0, t2 - t1, st["and_gates"], st["and_gates"] / st["seconds_total"], st["launches"],
        s.beta().tolist() == exp.tolist()), flush=True)
    s.close()
