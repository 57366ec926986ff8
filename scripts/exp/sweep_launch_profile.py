"""per-op launch profile of the merged 64-lambda sweep program (d = 100 CGD-15): serialised pass, diagnostic"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
OPS = "NOP MAC SUM SUBSUM IPMAC IPFIN IPMERGE MUL MULSUB ADD SUB ABS MAX DIV SQRT IDIVC CONST COPY REVEAL MAC2 MACK HDIFF EQ DIVB".split()
d, nl, it = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 15
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**60, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, 64, 56, "cgd", it, 0.0, 2, 1)
lam = np.linspace(0.001, 1.0, nl)
prog = lgc.Program(sysm, lambdas=lam)
L = prog.launches()
recs = np.frombuffer(prog.records().tobytes(), dtype=np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")]))
s = lgc.Solver(sysm, lambdas=lam); s.set_shares(shares); s.run(); t_run = s.stats()["seconds_total"]
s.set_shares(shares); s.run(profile=True)
g, e = s.profile(len(L))
agg = collections.OrderedDict()
for i, l in enumerate(L):
    key = OPS[recs[l["first_rec"]]["op"]]
    a = agg.setdefault(key, [0, 0, 0.0, 0.0, 0, 0])
    a[0] += 1; a[1] += l["nrec"]; a[2] += g[i]; a[3] += e[i]; a[4] += l["steps"]; a[5] += l["gates"]
print("sweep %d x d=%d cgd-%d: overlapped run %.3f s; serialised G %.3f E %.3f" % (nl, d, it, t_run, g.sum(), e.sum()))
print("%-8s %8s %9s %10s %10s %12s %10s %10s" % ("op", "launches", "records", "garble_ms", "eval_ms", "steps", "G Mgate/ms", "E Mgate/ms"))
for k, a in agg.items():
    print("%-8s %8d %9d %10.3f %10.3f %12d %10.2f %10.2f" % (k, a[0], a[1], a[2] * 1e3, a[3] * 1e3, a[4], a[5] / 1e6 / max(a[2] * 1e3, 1e-9), a[5] / 1e6 / max(a[3] * 1e3, 1e-9)))
