"""Where a level of the column-split kernel spends its cycles: runs a d = 20 Cholesky on a
GC_SPLIT_TRACE build (scripts/exp/build_variant.sh strace -DGC_SPLIT_TRACE=1) and reads the s_memtime stamps of record 0's
additions, glue wave 0 (hash q = 0) and hash wave 4 (q = 1).  Tags: 1 addition entered, 2 first AND hashed, 3 past a level
barrier, 4 operands rebuilt, 5 level hashed + result stored.  Usage: LGC_LIB=scripts/exp/libs/lib_strace.so python scripts/exp/split_trace.py"""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc

d, w, p = 20, 64, 56
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, w, p, "cholesky", 0, 0.0, 2, 0, 0, 0)
s = lgc.Solver(sysm); s.set_shares(shares); s.run()
L = lgc.lib()
for role in ("g", "e"):
    fn = getattr(L, "lgc_dbg_split_trace_" + role)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    fn(None, None, 1)
s.run()
for role in ("g", "e"):
    fn = getattr(L, "lgc_dbg_split_trace_" + role)
    buf = np.zeros(2 * 8192, dtype=np.uint64); n = np.zeros(2, dtype=np.uint32)
    assert fn(buf.ctypes.data, n.ctypes.data, 0) == 0
    for wv in (0, 1):
        st = buf[wv * 8192: wv * 8192 + int(n[wv])]
        tags = (st & np.uint64(15)).astype(int); t = (st >> np.uint64(4)).astype(np.int64)
        seg = collections.defaultdict(list)
        for i in range(1, len(t)):
            seg[(tags[i - 1], tags[i])].append(int(t[i] - t[i - 1]))
        print("== %s, wave %d (%s): %d stamps" % ("garbler" if role == "g" else "evaluator", 4 * wv, "glue + hash 0" if wv == 0 else "hash 1", len(t)))
        names = {(3, 4): "barrier passed -> operands rebuilt", (4, 5): "operands -> level hashed, result stored", (5, 3): "stored -> past the barrier",
                 (1, 2): "entered -> first AND hashed", (2, 3): "first AND stored -> past the barrier", (3, 1): "last level -> next addition entered (glue of a quotient bit, 2 hand-over barriers)",
                 (3, 3): "level without a gate of this wave",
                 (6, 7): "dual step: publish + post + first barrier", (7, 8): "dual step: operands read + hash + result stored", (8, 9): "dual step: second barrier",
                 (9, 10): "dual step: results read back", (10, 6): "dual step: glue of the multiplier row", (9, 7): "hash wave: barrier -> next descriptor decoded", (1, 4): "entered -> first AND's operand rebuilt", (4, 2): "first AND: operand -> hashed, result stored"}
        for k in sorted(seg):
            v = np.array(seg[k])
            print("   %-86s n %5d  median %6d  mean %7.0f  p90 %6d" % (names.get(k, str(k)), len(v), int(np.median(v)), v.mean(), int(np.percentile(v, 90))))
