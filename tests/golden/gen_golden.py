#!/usr/bin/env python3
"""Generate golden vectors for the fixed-point path with a pure-Python
big-integer twin of the reference's arithmetic (independent of oracle/*.c).

    python tests/golden/gen_golden.py        # rewrites tests/golden/*.json

The twin restates, with Python's unbounded ints and IEEE doubles:
  src/fixed.c:3-17, src/fixed.oc:78-248 (op semantics),
  src/linear.c:11-16,46-51, src/phase1.c:14-20,473-476,562-567,609-638,
  src/linear.oc:52-65, src/cgd.oc:96-203, src/cholesky.oc:51-87, src/ldlt.oc:50-90.
Its README known-answer (README.md:85-87) is asserted before anything is written.
Nothing from /root/reference is read except examples/readme_example.in, which
is also committed as tests/golden/readme_example.in (a data file).
"""
import json
import math
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def wrap(v, w):
    v &= (1 << w) - 1
    return v - (1 << w) if v >> (w - 1) else v


def d2f(x, p, w):                       # fixed.c:3-5 (C truncation)
    t = x * float(1 << p)
    lo, hi = -(1 << (w - 1)), (1 << (w - 1))
    if not (t > lo - 1 and t < hi):
        return lo                       # x86 cvttsd2si "integer indefinite"
    return int(t)


def f2d(f, p):                          # fixed.c:7-9
    return float(f) / float(1 << p)


def tdiv(a, b):                         # C integer division
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def mul(a, b, p, w): return wrap((a * b) >> p, w)


def div(a, b, p, w):
    if b == 0:
        return 1 if a < 0 else -1       # this build's definition (reference: unspecified)
    return wrap(tdiv(a << p, b), w)


def add(a, b, w): return wrap(a + b, w)
def sub(a, b, w): return wrap(a - b, w)
def fabs(a, w): return wrap(-a, w) if a < 0 else a


def cmp(a, b, w):
    if w == 64:                         # obig_cmp is unsigned
        a &= (1 << 64) - 1
        b &= (1 << 64) - 1
    return (a > b) - (a < b)


def ip(a, b, p, w):
    if w == 32:                         # wrapping int64 accumulator (fixed.oc:126-130)
        acc = 0
        for x, y in zip(a, b):
            acc = wrap(acc + x * y, 64)
        return wrap(acc >> p, 32)
    return wrap(sum(x * y for x, y in zip(a, b)) >> p, w)


def sqrt(a, p, w):
    if w == 32:                         # fixed.oc:228-240 verbatim semantics
        mask = (1 << (32 + p)) - 1
        x = ((a & 0xFFFFFFFFFFFFFFFF) << p) & mask
        r = 0
        e = mask + 1
        while e != 0:
            if (x & mask) >= ((r + e) & mask):
                x = (x - (r + e)) & 0xFFFFFFFFFFFFFFFF
                r = ((r >> 1) + e) & mask
            else:
                r >>= 1
            e >>= 2
        return wrap(r, 32)
    return wrap(math.isqrt((a & ((1 << 64) - 1)) << p), 64)


def idx(i, j):
    if j > i:
        i, j = j, i
    return i * (i + 1) // 2 + j


def quantize(vals, p1, n, w2):
    s = math.sqrt(math.pow(2, p1) * n)
    return [d2f(v / s, p1, w2) for v in vals]


def aggregate(Xq, yq, n, d, p1, w1):
    m = (1 << w1) - 1
    A = [0] * (d * (d + 1) // 2)
    b = [0] * d
    for i in range(d):
        for j in range(i + 1):
            if i == j:
                acc = 0.0
                for k in range(n):
                    v = f2d(Xq[k * d + i], p1)
                    acc += (v * v) * math.pow(2, p1)
                A[idx(i, j)] = d2f(acc / d, p1, w1) & m
            else:
                A[idx(i, j)] = sum(Xq[k * d + i] * Xq[k * d + j] for k in range(n)) & m
        b[i] = sum(Xq[k * d + i] * yq[k] for k in range(n)) & m
    return A, b


def circuit_input(A, b, d, lam, p, w):
    a = [wrap(v, w) for v in A]
    bb = [wrap(v, w) for v in b]
    lamq = d2f(lam, p, w)
    for i in range(d):
        for j in range(i + 1):
            ij = idx(i, j)
            a[ij] = add(a[ij], lamq, w) if i == j else wrap(tdiv(a[ij], d), w)
    bb = [wrap(tdiv(v, d), w) for v in bb]
    return a, bb


def cgd(a, b, d, p, w, iters):
    x = [0] * d; g = [0] * d; pv = [0] * d; gscl = [0] * d; pA = [0] * d
    ng = 0
    trace = []
    for i in range(d):
        g[i] = sub(g[i], b[i], w)
        t = fabs(g[i], w)
        if cmp(t, ng, w) > 0:
            ng = t
    for i in range(d):
        pv[i] = div(g[i], ng, p, w)
    for _ in range(iters):
        for i in range(d):
            acc = 0
            for j in range(d):
                acc = add(acc, mul(a[idx(i, j)], pv[j], p, w), w)
            pA[i] = acc
        q = ip(pA, pv, p, w)
        gp = ip(g, pv, p, w)
        eta = div(gp, q, p, w)
        ng = 0
        for i in range(d):
            x[i] = sub(x[i], mul(pv[i], eta, p, w), w)
            g[i] = sub(g[i], mul(eta, pA[i], p, w), w)
            t = fabs(g[i], w)
            if cmp(t, ng, w) > 0:
                ng = t
        for i in range(d):
            gscl[i] = div(g[i], ng, p, w)
        gAp = ip(pA, gscl, p, w)
        gamma = div(gAp, q, p, w)
        for i in range(d):
            pv[i] = sub(gscl[i], mul(pv[i], gamma, p, w), w)
        trace.append(list(x) + [gamma, eta, q, ng])
    return x, trace


def cholesky(a, b, d, p, w):
    a = list(a); b = list(b); y = [0] * d; beta = [0] * d
    for j in range(d):
        for k in range(j):
            for i in range(j, d):
                a[idx(i, j)] = sub(a[idx(i, j)], mul(a[idx(i, k)], a[idx(j, k)], p, w), w)
        a[idx(j, j)] = sqrt(a[idx(j, j)], p, w)
        for k in range(j + 1, d):
            a[idx(k, j)] = div(a[idx(k, j)], a[idx(j, j)], p, w)
    for i in range(d):
        for j in range(i):
            b[i] = sub(b[i], mul(a[idx(i, j)], y[j], p, w), w)
        y[i] = div(b[i], a[idx(i, i)], p, w)
    for i in range(d - 1, -1, -1):
        for j in range(d - 1, i, -1):
            y[i] = sub(y[i], mul(a[idx(j, i)], beta[j], p, w), w)
        beta[i] = div(y[i], a[idx(i, i)], p, w)
    return beta


def ldlt(a, b, d, p, w):
    a = list(a); b = list(b)
    for j in range(d):
        for k in range(j):
            t = mul(a[idx(j, k)], a[idx(k, k)], p, w)
            for i in range(j, d):
                a[idx(i, j)] = sub(a[idx(i, j)], mul(a[idx(i, k)], t, p, w), w)
        for k in range(j + 1, d):
            a[idx(k, j)] = div(a[idx(k, j)], a[idx(j, j)], p, w)
    for i in range(d):
        for j in range(i):
            b[i] = sub(b[i], mul(a[idx(i, j)], b[j], p, w), w)
    for i in range(d):
        b[i] = div(b[i], a[idx(i, i)], p, w)
    for i in range(d - 1, -1, -1):
        for j in range(d - 1, i, -1):
            b[i] = sub(b[i], mul(a[idx(j, i)], b[j], p, w), w)
    return b


def read_input(path):
    tok = open(path).read().split()
    n, d, P = int(tok[0]), int(tok[1]), int(tok[2])
    pos = 3 + 2
    start = []
    for _ in range(P):
        start.append(int(tok[pos + 1])); pos += 2
    assert int(tok[pos]) == n and int(tok[pos + 1]) == d
    pos += 2
    X = [float(t) for t in tok[pos:pos + n * d]]; pos += n * d
    assert int(tok[pos]) == n
    pos += 1
    y = [float(t) for t in tok[pos:pos + n]]
    return n, d, P, start, X, y


def pipeline(n, d, X, y, p1, p2, w, alg, iters, lam):
    Xq = quantize(X, p1, n, w); yq = quantize(y, p1, n, w)
    A, b = aggregate(Xq, yq, n, d, p1, w)
    a, bb = circuit_input(A, b, d, lam, p2, w)
    if alg == "cgd":
        return cgd(a, bb, d, p2, w, iters)[0], (Xq, yq, A, b, a, bb)
    if alg == "cholesky":
        return cholesky(a, bb, d, p2, w), (Xq, yq, A, b, a, bb)
    return ldlt(a, bb, d, p2, w), (Xq, yq, A, b, a, bb)


def rnd_word(rng, w, kind):
    if kind == 0:
        return wrap(rng.getrandbits(w), w)
    if kind == 1:                      # small magnitudes
        return wrap(rng.getrandbits(rng.randint(1, w - 1)) * rng.choice((1, -1)), w)
    return rng.choice([0, 1, -1, (1 << (w - 1)) - 1, -(1 << (w - 1)), 1 << (w - 2), -(1 << (w - 2))])


def gen_ops(rng):
    cases = []
    for w, precs in ((64, (56, 54, 30, 63, 0, 1)), (32, (30, 20, 31, 0, 1, 15))):
        for p in precs:
            for _ in range(60):
                kind = rng.randint(0, 2)
                a = rnd_word(rng, w, kind); b = rnd_word(rng, w, rng.randint(0, 2))
                cases.append(dict(w=w, p=p, a=a, b=b, add=add(a, b, w), sub=sub(a, b, w),
                                  abs=fabs(a, w), cmp=cmp(a, b, w), mul=mul(a, b, p, w),
                                  div=div(a, b, p, w), sqrt=sqrt(a, p, w)))
            for _ in range(8):
                n = rng.randint(1, 40)
                va = [rnd_word(rng, w, rng.randint(0, 1)) for _ in range(n)]
                vb = [rnd_word(rng, w, rng.randint(0, 1)) for _ in range(n)]
                cases.append(dict(w=w, p=p, va=va, vb=vb, ip=ip(va, vb, p, w)))
    return cases


def gen_system(rng, n, d, sigma=0.1):
    # experiments/generate_tests.py:159-169 (distribution only)
    cols = [[rng.gauss(0, 1) for _ in range(n)] for _ in range(d)]
    cols = [[v / max(abs(u) for u in c) for v in c] for c in cols]
    beta = [rng.random() for _ in range(d)]
    X = [cols[j][k] for k in range(n) for j in range(d)]
    y = [sum(cols[j][k] * beta[j] for j in range(d)) + rng.gauss(0, sigma) for k in range(n)]
    return X, y


def main():
    rng = random.Random(20261002)
    # --- pin the twin itself on the README known answer (README.md:85-87)
    n, d, P, start, X, y = read_input(os.path.join(HERE, "readme_example.in"))
    beta, inter = pipeline(n, d, X, y, 56, 56, 64, "cgd", 10, 0.001)
    printed = " ".join("%20.15f" % f2d(v, 56) for v in beta).split()
    assert printed == ["0.984331027786964", "0.792399824970372", "0.754117840176144",
                       "0.592849130685193", "0.057351715952213"], printed
    readme = dict(n=n, d=d, P=P, start=start, p=56, w=64, iters=10, lam=0.001,
                  Xq=inter[0], yq=inter[1], A=inter[2], b=inter[3], a=inter[4], bb=inter[5],
                  beta_cgd=beta,
                  beta_cholesky=pipeline(n, d, X, y, 56, 56, 64, "cholesky", 0, 0.001)[0],
                  beta_ldlt=pipeline(n, d, X, y, 56, 56, 64, "ldlt", 0, 0.001)[0],
                  trace=cgd(inter[4], inter[5], d, 56, 64, 10)[1],
                  printed=printed)
    json.dump(readme, open(os.path.join(HERE, "readme_kat.json"), "w"))
    json.dump(gen_ops(rng), open(os.path.join(HERE, "ops.json"), "w"))
    systems = []
    for (n, d, w, p, iters) in ((50, 4, 64, 56, 8), (200, 8, 64, 54, 12), (120, 6, 32, 30, 6),
                                (300, 10, 32, 28, 10), (64, 3, 64, 40, 5)):
        X, y = gen_system(rng, n, d)
        lam = 0.001
        out = {}
        for alg in ("cgd", "cholesky", "ldlt"):
            out[alg], inter = pipeline(n, d, X, y, p, p, w, alg, iters, lam)
        systems.append(dict(n=n, d=d, w=w, p=p, iters=iters, lam=lam, X=X, y=y,
                            Xq=inter[0], yq=inter[1], A=inter[2], b=inter[3],
                            a=inter[4], bb=inter[5], beta=out))
    json.dump(systems, open(os.path.join(HERE, "systems.json"), "w"))
    print("golden vectors written")


if __name__ == "__main__":
    main()
