import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linreg_gc as lgc, orc
from helpers import oracle_solve, split_shares, synth_system
oracle = orc.load()
for (d, w, p, alg, it) in ((1000, 64, 56, "cgd", 3), (1000, 32, 30, "cgd", 3), (800, 64, 56, "ldlt", 0)):
    rng = np.random.default_rng(d + w)
    A, b = synth_system(oracle, rng, 2 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, it, 1e-3, 2, 1, 0, 1 if alg == "cgd" else 0)
    t0 = time.time(); s = lgc.Solver(sysm); s.set_shares(shares); t1 = time.time(); s.run(); t2 = time.time()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, it, 1e-3, 1, trace=(alg == "cgd"))
    st = s.stats()
    ok = (s.trace().tolist() == exp[1].tolist() and s.beta().tolist() == exp[0].tolist()) if alg == "cgd" else s.beta().tolist() == exp.tolist()
    print("d=%d w=%d %s-%d: build %.1fs run %.2fs gates %.3e (%.3e AND/s) launches %d exact=%s" % (d, w, alg, it, t1 - t0, t2 - t1, st["and_gates"], st["and_gates"] / st["seconds_total"], st["launches"], ok), flush=True)
    s.close()
