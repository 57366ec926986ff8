"""Randomised parity sweep on the GPU (not part of the pytest suite; run through gpurun):
random dimension, algorithm, width, precision, share count, input path, lambda and seed; every
result (beta, trace, revealed inputs) is compared bit for bit with the CPU oracle.
    python tests/tools/gpu_fuzz.py [cases] [seed] [big|small|huge]     (big: dimensions that reach the MAC and wide kernels;
    huge: matrix-vector launches of several rounds -- the MAC kernels' record queue with chunks of many records per wave)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linreg_gc as lgc, orc
from helpers import oracle_solve, split_shares, sx, synth_system

oracle = orc.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
BIG = len(sys.argv) > 3 and sys.argv[3] in ("big", "huge")
HUGE = len(sys.argv) > 3 and sys.argv[3] == "huge"
t0 = time.time()
bad = 0
for c in range(cases):
    w = int(rng.choice([32, 64]))
    p = int(rng.integers(8, 30)) if w == 32 else int(rng.integers(20, 58))
    alg = str(rng.choice(["cgd", "cholesky", "ldlt"]))
    d = int(rng.integers(1, 48)) if alg == "cgd" else int(rng.integers(1, 28))
    if BIG:
        d = int(rng.integers(48, 220)) if alg == "cgd" else int(rng.integers(28, 72))
    if HUGE:
        d = int(rng.integers(220, 420)) if alg == "cgd" else int(rng.integers(72, 130))
    n = int(rng.integers(d + 2, 4 * d + 10))
    normalize = int(rng.integers(0, 2))
    nsh = int(rng.integers(1, 6)) if normalize else 2
    iters = int(rng.integers(1, 3 if BIG else 8))
    lam = float(rng.choice([0.0, 1e-6, 1e-3, 0.1, 1.0]))
    sweep = normalize and rng.random() < 0.25
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, nsh, w)
    seed = bytes(rng.integers(0, 256, size=16, dtype=np.uint8))
    tag = "case %d: w=%d p=%d %s d=%d n=%d normalize=%d nshares=%d iters=%d lam=%g sweep=%s" % (c, w, p, alg, d, n, normalize, nsh, iters, lam, sweep)
    try:
        if sweep:
            lams = [lam, 0.5 * lam + 1e-4, 0.25]
            s = lgc.Solver(lgc.make_system(d, w, p, alg, iters, 0.0, nsh, 1, 0, 0), seed=seed, lambdas=lams)
            s.set_shares(shares); s.run()
            got = s.beta()
            for t, l in enumerate(lams):
                exp = oracle_solve(oracle, A, b, d, w, p, alg, iters, l, 1)[0]
                assert got[t].tolist() == exp.tolist(), "beta (sweep %d)" % t
        else:
            trace = int(alg == "cgd")
            s = lgc.Solver(lgc.make_system(d, w, p, alg, iters, lam, nsh, normalize, 1, trace), seed=seed)
            s.set_shares(shares); s.run(profile=bool(rng.integers(0, 2)))
            exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=bool(trace))
            if trace:
                exp, tr = exp
                assert s.trace().tolist() == np.asarray(tr).reshape(iters, d + 4).tolist(), "trace"
            assert s.inputs().tolist() == np.concatenate([a, bb]).tolist(), "inputs"
            assert s.beta().tolist() == np.asarray(exp).tolist(), "beta"
        s.close()
    except AssertionError as e:
        bad += 1
        print("MISMATCH", tag, e, flush=True)
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
