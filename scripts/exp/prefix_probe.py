"""the first launches of a normalised program (share sums, normalizer, mirror copies), launch by launch (diagnostic)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
OPS = "NOP MAC SUM SUBSUM IPMAC IPFIN IPMERGE MUL MULSUB ADD SUB ABS MAX DIV SQRT IDIVC CONST COPY REVEAL MAC2 MACK HDIFF EQ DIVB".split()
for d, w, p, nsh in ((500, 32, 30, 5), (500, 64, 56, 2), (100, 64, 56, 2)):
    rng = np.random.default_rng(0)
    T = d * (d + 1) // 2
    shares = rng.integers(0, 2**(w - 2), size=(nsh, T + d), dtype=np.uint64)
    sysm = lgc.make_system(d, w, p, "cgd", 1, 0.001, nsh, 1, 0, 0)
    prog = lgc.Program(sysm)
    L = prog.launches()
    recs = np.frombuffer(prog.records().tobytes(), dtype=np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")]))
    s = lgc.Solver(sysm); s.set_shares(shares); s.run(profile=True); s.run(profile=True)
    g, e = s.profile(len(L))
    print("d=%d w=%d shares=%d normalised: %d launches" % (d, w, nsh, len(L)))
    for i, l in enumerate(L[:8]):
        print("  launch %d %-6s records %7d steps %9d table %7.2f GB  garble %7.3f ms  eval %7.3f ms" % (
            i, OPS[recs[l["first_rec"]]["op"]], l["nrec"], l["steps"], l["steps"] * 2048 / 1e9, g[i] * 1e3, e[i] * 1e3))
    s.close()
