"""per-launch kernel time breakdown by record type (diagnostic, not a test)"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
OPS = "NOP MAC SUM SUBSUM IPMAC IPFIN IPMERGE MUL MULSUB ADD SUB ABS MAX DIV SQRT IDIVC CONST COPY REVEAL MAC2 MACK HDIFF EQ DIVB".split()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 500
alg = sys.argv[2] if len(sys.argv) > 2 else "cgd"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = int(sys.argv[4]) if len(sys.argv) > 4 else 64
p = 56 if w == 64 else 30
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
prog = lgc.Program(sysm)
L = prog.launches()
recs = np.frombuffer(prog.records().tobytes(), dtype=np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")]))
s = lgc.Solver(sysm); s.set_shares(shares); s.run(profile=True); s.run(profile=True)
g, e = s.profile(len(L))
agg = collections.OrderedDict()
for i, l in enumerate(L):
    r0 = recs[l["first_rec"]]
    key = OPS[r0["op"]]
    a = agg.setdefault(key, [0, 0, 0.0, 0.0, 0, 0])
    a[0] += 1; a[1] += l["nrec"]; a[2] += g[i]; a[3] += e[i]; a[4] += l["steps"]; a[5] = max(a[5], int(recs["cnt"][l["first_rec"]:l["first_rec"]+l["nrec"]].max()))
print("d=%d %s-%d w=%d total %.4fs (G %.4f E %.4f)" % (d, alg, iters, w, s.stats()["seconds_total"], g.sum(), e.sum()))
print("%-8s %8s %9s %10s %10s %12s %6s" % ("op", "launches", "records", "garble_ms", "eval_ms", "steps", "maxcnt"))
for k, a in agg.items():
    print("%-8s %8d %9d %10.3f %10.3f %12d %6d" % (k, a[0], a[1], a[2] * 1e3, a[3] * 1e3, a[4], a[5]))
