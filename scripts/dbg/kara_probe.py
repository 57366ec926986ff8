"""debug probe: Karatsuba path on the GPU against the oracle at several d"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import linreg_gc as lgc
import orc
from helpers import oracle_solve, split_shares, synth_system
oracle = orc.load()
for d, it in ((130, 2), (160, 2), (200, 2), (260, 2), (300, 2), (400, 1), (500, 1)):
    rng = np.random.default_rng(d)
    A, b = synth_system(oracle, rng, 4 * d, d, 64, 56)
    shares = split_shares(rng, A, b, 2, 64)
    exp, _, _ = oracle_solve(oracle, A, b, d, 64, 56, "cgd", it, 0.0, 0, trace=True)
    for kara in (1, 0):
        lgc.set_karatsuba(kara)
        sysm = lgc.make_system(d, 64, 56, "cgd", it, 0.0, 2, 0, 0, 1)
        P = lgc.Program(sysm)
        macs = [(L["nrec"], L["steps"]) for L in P.launches() if L["mac_only"]]
        hd = sorted(set(L["nrec"] for L in P.launches() if not L["mac_only"] and L["nrec"] in (d, d * (d + 1) // 2)))
        s = lgc.Solver(sysm, seed=bytes(range(16)))
        s.set_shares(shares)
        t0 = time.time(); s.run(); dt = time.time() - t0
        ok = s.beta().tolist() == exp[0].tolist()
        tr = s.trace()
        bad = [i for i in range(it) if tr[i].tolist() != exp[1][i].tolist()]
        print("d", d, "kara", kara, "exact", ok, "bad iterations", bad, "gates %.3e" % P.info.total_gates, "mac launches", macs[:2], len(macs), "%.3f s" % dt, flush=True)
        if bad:
            i = bad[0]
            diff = [j for j in range(d + 4) if tr[i][j] != exp[1][i][j]]
            print("   first bad iteration", i, "differing entries", len(diff), diff[:8])
        s.close()
