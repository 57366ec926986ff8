"""shared helpers for the parity tests"""
import numpy as np


def synth_system(oracle, rng, n, d, w, p, lam=0.001, sigma=0.1):
    """experiments/generate_tests.py:159-169 distribution, quantised and aggregated
    by the oracle.  Returns (A_total, b_total) as uint64 (T and d entries)."""
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    beta = rng.random(d)
    y = X @ beta + sigma * rng.standard_normal(n)
    Xq = oracle.quantize(X, p, n, w); yq = oracle.quantize(y, p, n, w)
    A, b = oracle.aggregate(Xq, yq, n, d, p, w)
    return A, b


def split_shares(rng, A, b, nshares, w):
    """additive shares mod 2^w of the (T + d) vector; share-major (nshares, T + d)"""
    tot = np.concatenate([A, b]).astype(np.uint64)
    m = np.uint64((1 << w) - 1) if w < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sh = rng.integers(0, 2 ** 63, size=(nshares, tot.size), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(nshares, tot.size), dtype=np.uint64)
    sh &= m
    with np.errstate(over="ignore"):
        sh[0] = (tot - sh[1:].sum(axis=0, dtype=np.uint64)) & m
    return sh


def oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=False):
    a = oracle.sum_shares(np.asarray(A, dtype=np.uint64)[None, :], w)
    bb = oracle.sum_shares(np.asarray(b, dtype=np.uint64)[None, :], w)
    if normalize:
        a, bb = oracle.circuit_input(a, bb, d, lam, p, w)
    if alg == "cgd":
        return oracle.cgd(a, bb, d, p, w, iters, trace=trace), a, bb
    if alg == "cholesky":
        return oracle.cholesky(a, bb, d, p, w), a, bb
    return oracle.ldlt(a, bb, d, p, w), a, bb


def sx(v, w):
    v = np.asarray(v, dtype=np.uint64)
    if w == 32:
        return v.astype(np.uint32).astype(np.int32).astype(np.int64)
    return v.astype(np.int64)
