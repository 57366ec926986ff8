"""ctypes binding of the CPU semantic oracle (oracle/liblinreg_oracle.so).

Test infrastructure only.  Builds the library with oracle/Makefile on demand.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Input(C.Structure):
    _fields_ = [("n", C.c_size_t), ("d", C.c_size_t), ("P", C.c_size_t),
                ("start", C.POINTER(C.c_size_t)), ("X", C.POINTER(C.c_double)),
                ("y", C.POINTER(C.c_double))]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = L = lib
        i64, dbl, sz, vp, ci = C.c_int64, C.c_double, C.c_size_t, C.c_void_p, C.c_int
        for name, res, args in [
            ("orc_wrap", i64, [i64, ci]),
            ("orc_double_to_fixed", i64, [dbl, ci, ci]),
            ("orc_fixed_to_double", dbl, [i64, ci]),
            ("orc_add", i64, [i64, i64, ci]), ("orc_sub", i64, [i64, i64, ci]),
            ("orc_abs", i64, [i64, ci]), ("orc_cmp", ci, [i64, i64, ci]),
            ("orc_mul", i64, [i64, i64, ci, ci]), ("orc_div", i64, [i64, i64, ci, ci]),
            ("orc_sqrt", i64, [i64, ci, ci]),
            ("orc_inner_product", i64, [vp, vp, sz, ci, ci]),
            ("orc_idx", sz, [sz, sz]),
            ("orc_quantize", None, [vp, sz, ci, sz, ci, vp]),
            ("orc_aggregate", None, [vp, vp, sz, sz, ci, ci, vp, vp]),
            ("orc_phase1_ti_shares", ci, [vp, vp, sz, sz, ci, ci, sz, vp, vp, vp, vp, vp]),
            ("orc_phase1_ot_shares", ci, [vp, vp, sz, sz, ci, ci, sz, vp, vp, vp, vp, vp]),
            ("orc_convert_shares", None, [vp, sz, ci, ci, ci, ci, vp]),
            ("orc_sum_shares", None, [vp, sz, sz, ci, vp]),
            ("orc_circuit_input", None, [vp, vp, sz, dbl, ci, ci]),
            ("orc_cgd", None, [vp, vp, sz, ci, ci, ci, vp, vp]),
            ("orc_cholesky", None, [vp, vp, sz, ci, ci, vp]),
            ("orc_ldlt", None, [vp, vp, sz, ci, ci, vp]),
            ("orc_read_input", ci, [C.c_char_p, vp]),
            ("orc_free_input", None, [vp]),
            ("orc_linreg", ci, [vp, ci, ci, ci, ci, ci, ci, dbl, vp]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args

    # ---- scalars
    def mul(self, a, b, p, w): return self.lib.orc_mul(a, b, p, w)
    def div(self, a, b, p, w): return self.lib.orc_div(a, b, p, w)
    def sqrt(self, a, p, w): return self.lib.orc_sqrt(a, p, w)
    def add(self, a, b, w): return self.lib.orc_add(a, b, w)
    def sub(self, a, b, w): return self.lib.orc_sub(a, b, w)
    def abs(self, a, w): return self.lib.orc_abs(a, w)
    def cmp(self, a, b, w): return self.lib.orc_cmp(a, b, w)
    def wrap(self, a, w): return self.lib.orc_wrap(a, w)

    def inner_product(self, a, b, p, w):
        a = np.ascontiguousarray(a, dtype=np.int64); b = np.ascontiguousarray(b, dtype=np.int64)
        return self.lib.orc_inner_product(_p(a), _p(b), len(a), p, w)

    # ---- pipeline
    def quantize(self, vals, p1, n, w2):
        vals = np.ascontiguousarray(vals, dtype=np.float64).ravel()
        out = np.empty(vals.size, dtype=np.int64)
        self.lib.orc_quantize(_p(vals), vals.size, p1, n, w2, _p(out))
        return out

    def aggregate(self, Xq, yq, n, d, p1, w1):
        Xq = np.ascontiguousarray(Xq, dtype=np.int64); yq = np.ascontiguousarray(yq, dtype=np.int64)
        A = np.empty(d * (d + 1) // 2, dtype=np.uint64); b = np.empty(d, dtype=np.uint64)
        self.lib.orc_aggregate(_p(Xq), _p(yq), n, d, p1, w1, _p(A), _p(b))
        return A, b

    def _shares(self, fn, Xq, yq, n, d, p1, w1, start, rnd):
        Xq = np.ascontiguousarray(Xq, dtype=np.int64); yq = np.ascontiguousarray(yq, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.uint64); P = len(start)
        rnd = np.ascontiguousarray(rnd, dtype=np.uint64)
        T = d * (d + 1) // 2
        sA = np.zeros((P, T), dtype=np.uint64); sb = np.zeros((P, d), dtype=np.uint64)
        used = C.c_size_t(rnd.size)
        rc = fn(_p(Xq), _p(yq), n, d, p1, w1, P, _p(start), _p(rnd), C.byref(used), _p(sA), _p(sb))
        if rc:
            raise RuntimeError("oracle share simulation failed rc=%d" % rc)
        return sA, sb, used.value

    def ti_shares(self, *a): return self._shares(self.lib.orc_phase1_ti_shares, *a)

    def ti_shares_stream(self, Xq, yq, n, d, p1, w1, start, words_fn):
        """as ti_shares, the TI's stream supplied pair by pair: words_fn(first_word, count) -> uint64 array"""
        Xq = np.ascontiguousarray(Xq, dtype=np.int64); yq = np.ascontiguousarray(yq, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.uint64); P = len(start)
        T = d * (d + 1) // 2
        sA = np.zeros((P, T), dtype=np.uint64); sb = np.zeros((P, d), dtype=np.uint64)
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64))

        def cb(_ctx, first, count, out):
            w = np.ascontiguousarray(words_fn(first, count), dtype=np.uint64)
            C.memmove(out, w.ctypes.data, count * 8)
            return 0
        used = C.c_size_t(0)
        fn = self.lib.orc_phase1_ti_shares_cb
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_void_p, CB, C.c_void_p,
                       C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p]
        rc = fn(_p(Xq), _p(yq), n, d, p1, w1, P, _p(start), CB(cb), None, C.byref(used), _p(sA), _p(sb))
        if rc:
            raise RuntimeError("oracle share simulation failed rc=%d" % rc)
        return sA, sb, used.value
    def ot_shares(self, *a): return self._shares(self.lib.orc_phase1_ot_shares, *a)

    def convert_shares(self, s, p1, p2, w1, w2):
        s = np.ascontiguousarray(s, dtype=np.uint64); out = np.empty_like(s)
        self.lib.orc_convert_shares(_p(s), s.size, p1, p2, w1, w2, _p(out))
        return out

    def sum_shares(self, shares, w2):
        shares = np.ascontiguousarray(shares, dtype=np.uint64)
        P, count = shares.shape
        out = np.empty(count, dtype=np.int64)
        self.lib.orc_sum_shares(_p(shares), P, count, w2, _p(out))
        return out

    def circuit_input(self, a, b, d, lam, p, w):
        a = np.array(a, dtype=np.int64); b = np.array(b, dtype=np.int64)
        self.lib.orc_circuit_input(_p(a), _p(b), d, lam, p, w)
        return a, b

    def cgd(self, a, b, d, p, w, iters, trace=False):
        a = np.ascontiguousarray(a, dtype=np.int64); b = np.ascontiguousarray(b, dtype=np.int64)
        beta = np.zeros(d, dtype=np.int64)
        tr = np.zeros((iters, d + 4), dtype=np.int64) if trace else None
        self.lib.orc_cgd(_p(a), _p(b), d, p, w, iters, _p(beta), _p(tr) if trace else None)
        return (beta, tr) if trace else beta

    def cholesky(self, a, b, d, p, w):
        a = np.ascontiguousarray(a, dtype=np.int64); b = np.ascontiguousarray(b, dtype=np.int64)
        beta = np.zeros(d, dtype=np.int64)
        self.lib.orc_cholesky(_p(a), _p(b), d, p, w, _p(beta))
        return beta

    def ldlt(self, a, b, d, p, w):
        a = np.ascontiguousarray(a, dtype=np.int64); b = np.ascontiguousarray(b, dtype=np.int64)
        beta = np.zeros(d, dtype=np.int64)
        self.lib.orc_ldlt(_p(a), _p(b), d, p, w, _p(beta))
        return beta

    def read_input(self, path):
        inp = Input()
        rc = self.lib.orc_read_input(path.encode(), C.byref(inp))
        if rc:
            raise RuntimeError("orc_read_input rc=%d" % rc)
        n, d, P = inp.n, inp.d, inp.P
        res = dict(n=n, d=d, P=P, start=[inp.start[i] for i in range(P)],
                   X=np.ctypeslib.as_array(inp.X, (n * d,)).reshape(n, d).copy(),
                   y=np.ctypeslib.as_array(inp.y, (n,)).copy())
        self.lib.orc_free_input(C.byref(inp))
        return res

    def linreg_file(self, path, p1, p2, w1, w2, alg, iters, lam):
        inp = Input()
        rc = self.lib.orc_read_input(path.encode(), C.byref(inp))
        if rc:
            raise RuntimeError("orc_read_input rc=%d" % rc)
        beta = np.zeros(inp.d, dtype=np.int64)
        self.lib.orc_linreg(C.byref(inp), p1, p2, w1, w2, alg, iters, lam, _p(beta))
        self.lib.orc_free_input(C.byref(inp))
        return beta


def load():
    so = os.path.join(ODIR, "liblinreg_oracle.so")
    src = [os.path.join(ODIR, f) for f in ("linreg_oracle.c", "linreg_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ODIR, "liblinreg_oracle.so"], stdout=sys.stderr)
    return Oracle(C.CDLL(so))
