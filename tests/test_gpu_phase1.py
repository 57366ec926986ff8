"""GPU parity tests for the phase-1 aggregation arithmetic (src/phase1.c) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(oracle, rng, n, d, p, w):
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    y = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
    return oracle.quantize(X, p, n, w).reshape(n, d), oracle.quantize(y, p, n, w)


@pytest.mark.parametrize("n,d,p,w", [(300, 7, 56, 64), (1000, 70, 56, 64), (257, 9, 30, 32), (5000, 130, 54, 64)])
def test_local_block_matches_oracle(lgc, oracle, n, d, p, w):
    rng = np.random.default_rng(n + d)
    Xq, yq = _data(oracle, rng, n, d, p, w)
    A, b = oracle.aggregate(Xq, yq, n, d, p, w)          # totals: a single DP owning everything
    ph = lgc.Phase1(Xq, yq, w, p)
    gA, gb = ph.local(0, d, with_y=True)
    assert gA.tolist() == [int(v) for v in A]
    assert gb.tolist() == [int(v) for v in b]
    # a middle block without y
    c0, c1 = d // 3, d - 1
    blk = ph.local(c0, c1)
    exp = [int(A[oracle.lib.orc_idx(c0 + i, c0 + j)]) for i in range(c1 - c0) for j in range(i + 1)]
    assert blk.tolist() == exp
    ph.close()


def test_ti_mode_shares_match_oracle(lgc, oracle, gccpu):
    """whole TI-mode phase 1 for P data providers: GPU arithmetic + GPU TI stream vs the oracle's
    share-level simulation fed with the same (CPU-recomputed) AES-CTR stream"""
    rng = np.random.default_rng(3)
    n, d, p, w = 64, 6, 56, 64
    start = [0, 2, 4]
    P = len(start)
    Xq, yq = _data(oracle, rng, n, d, p, w)
    seed = bytes(range(16, 32))
    own = lambda row: max(k for k in range(P) if start[k] <= min(row, d - 1) or k == 0) if row < d else P - 1
    owner = [own(r) for r in range(d + 1)]
    pairs = [(i, j) for i in range(d + 1) for j in range(min(i, d - 1) + 1)
             if i != j and owner[i] != owner[j]]
    x, y, r, xyr = lgc.ti_generate(seed, 0, len(pairs), n, w)
    # the same stream on the CPU (AES-NI), consumed x, y, r per pair
    words = gccpu.ti_stream_words(seed, 0, len(pairs) * (2 * n + 1), w)
    assert np.array_equal(words.reshape(len(pairs), 2 * n + 1)[:, :n], x)
    assert np.array_equal(words.reshape(len(pairs), 2 * n + 1)[:, n:2 * n], y)
    assert np.array_equal(words.reshape(len(pairs), 2 * n + 1)[:, 2 * n], r)
    sA, sb, used = oracle.ti_shares(Xq, yq, n, d, p, w, start, words)
    assert used == len(pairs) * (2 * n + 1)
    T = d * (d + 1) // 2
    gA = np.zeros((P, T), dtype=np.uint64); gb = np.zeros((P, d), dtype=np.uint64)
    dps = []
    for k in range(P):
        c0 = start[k]; c1 = start[k + 1] if k + 1 < P else d
        Xk = np.zeros_like(Xq); Xk[:, c0:c1] = Xq[:, c0:c1]          # a DP only holds its own columns
        dps.append(lgc.Phase1(Xk, yq if k == P - 1 else None, w, p))
        if k == P - 1:
            blk, bb = dps[k].local(c0, c1, with_y=True)
            gb[k, c0:c1] = bb
        else:
            blk = dps[k].local(c0, c1)
        for i in range(c1 - c0):
            for j in range(i + 1):
                gA[k, oracle.lib.orc_idx(c0 + i, c0 + j)] = blk[i * (i + 1) // 2 + j]
    for q, (i, j) in enumerate(pairs):
        a, b = owner[i], owner[j]          # a owns row i (gets y, xy - r); b owns row j (gets x, r)
        bx = dps[b].mask([j], x[q:q + 1], +1)                       # b + x -> a
        ay = dps[a].mask([i], y[q:q + 1], -1)                       # a - y -> b
        share_a = dps[a].dot(bx, B=y[q:q + 1], sub=xyr[q:q + 1])[0]  # <b+x, y> - (xy - r)
        ay2, share_a2 = dps[a].ti_a(i, y[q], bx[0], xyr[q])          # the same two results in one pass
        assert np.array_equal(ay2, ay[0]) and int(share_a2) == int(share_a)
        share_b = dps[b].dot(ay, cols=[j], sub=r[q:q + 1])[0]        # <a-y, b> - r
        if i < d:
            gA[a, oracle.lib.orc_idx(i, j)] = share_a; gA[b, oracle.lib.orc_idx(i, j)] = share_b
        else:
            gb[a, j] = share_a; gb[b, j] = share_b
    assert np.array_equal(gA, sA)
    assert np.array_equal(gb, sb)
    A, bt = oracle.aggregate(Xq, yq, n, d, p, w)
    assert [int(v) for v in gA.sum(axis=0)] == [int(v) for v in A]
    for h in dps:
        h.close()
