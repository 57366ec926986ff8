"""diagnostic: mid-size systems with every pairing of the 16-wave and 4-wave kernels, bit-exact vs the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
import linreg_gc as lgc
import orc
from helpers import oracle_solve, split_shares, synth_system
oracle = orc.load()
bad = 0
for d, alg, w, p in [(60, "cholesky", 64, 56), (48, "cgd", 64, 56), (40, "ldlt", 32, 30), (270, "cholesky", 64, 56)]:
    rng = np.random.default_rng(d)
    A, b = synth_system(oracle, rng, 4 * d, d, w, p)
    sh = split_shares(rng, A, b, 2, w)
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, 6, 0.0, 0)
    for g, e in [(1, 1), (0, 0), (1, 0), (0, 1)]:
        lgc.set_split_kernels(g, e)
        s = lgc.Solver(lgc.make_system(d, w, p, alg, 6, 0.0, 2, 0, 0, 0), seed=bytes(range(16)))
        s.set_shares(sh); s.run()
        ok = s.beta().tolist() == (exp[0] if alg == "cgd" and isinstance(exp, tuple) else exp).tolist()
        print("d=%d %s w=%d split(g=%d,e=%d): %s  %.3f s" % (d, alg, w, g, e, "exact" if ok else "MISMATCH", s.stats()["seconds_total"]), flush=True)
        bad += not ok
        s.close()
lgc.set_split_kernels(1, 1)
print("mismatches:", bad)
