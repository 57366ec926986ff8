"""Op-level checks of the wave-shaped circuits (gc_circuits.h, plaintext backend) against the
semantic oracle: random and edge operands for every word operation, every width / precision the
command line accepts (src/cmd/linreg.c:85-88), in both step orders the GPU kernels use."""
import numpy as np
import pytest

OP = dict(MUL=7, ADD=9, SUB=10, ABS=11, DIV=13, SQRT=14, IDIVC=15, DIVB=23)


def _operands(rng, w, n):
    m = (1 << w) - 1
    edge = [0, 1, 2, 3, m, m - 1, 1 << (w - 1), (1 << (w - 1)) - 1, (1 << (w - 1)) + 1, 5, 0x5555555555555555 & m,
            0xAAAAAAAAAAAAAAAA & m, 1 << (w // 2), (1 << (w // 2)) - 1]
    a = [x for x in edge for _ in edge]
    b = [y for _ in edge for y in edge]
    r = rng.integers(0, 1 << 63, size=(2, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(2, n), dtype=np.uint64)
    # mixed magnitudes: shift random values right by random amounts, keep signs varied
    sh = rng.integers(0, w, size=(2, n)).astype(np.uint64)
    r = (r & np.uint64(m)) >> sh
    neg = rng.integers(0, 2, size=(2, n)).astype(bool)
    r = np.where(neg, (~r + np.uint64(1)) & np.uint64(m), r)
    a = np.concatenate([np.array(a, dtype=np.uint64), r[0]])
    b = np.concatenate([np.array(b, dtype=np.uint64), r[1]])
    return a, b


def _signed(v, w):
    v = int(v) & ((1 << w) - 1)
    return v - (1 << w) if v >> (w - 1) else v


PRECS = {64: [0, 1, 30, 56, 60, 61, 62, 63], 32: [0, 1, 15, 28, 30, 31]}


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("w", [64, 32])
def test_div_matches_oracle(gccpu, oracle, w, paired):
    rng = np.random.default_rng(100 + w)
    for p in PRECS[w]:
        a, b = _operands(rng, w, 150)
        got, steps = gccpu.plain_op(OP["DIV"], w, p, a, b, paired=paired)
        for x, y, g in zip(a, b, got):
            sx_, sy = _signed(x, w), _signed(y, w)
            exp = oracle.div(sx_, sy, p, w) & ((1 << w) - 1)
            assert int(g) == exp, (w, p, hex(int(x)), hex(int(y)), hex(int(g)), hex(exp))
        # 7 dependent levels / 12 gate steps per quotient bit + three conditional negates + zero detector
        assert steps <= 12 * (w + p) + 3 * 12 + 8


@pytest.mark.parametrize("p", [0, 1, 30, 56, 63])
def test_bounded_division_is_the_division_when_the_dividend_is_no_larger(gccpu, oracle, p):
    """OP_DIVB (CGD's g_i / max|g|, 64 bit): p + 1 quotient bits instead of 64 + p -- equal to tdiv(a << p, b), and to the
    full divider, for every |a| <= |b|: random pairs, equal magnitudes of either sign, zero dividend, the divisor 1,
    INT_MIN as divisor and as both, neighbours of the divisor, and 0 / 0 (the zero detector's all-ones magnitude)"""
    w = 64
    rng = np.random.default_rng(500 + p)
    mask = (1 << w) - 1
    top = 1 << 63
    a, b = [], []
    for _ in range(300):
        y = int(rng.integers(1, 1 << 62)) >> int(rng.integers(0, 60))
        y = max(y, 1)
        x = int(rng.integers(0, y + 1))
        for sx_ in (1, -1):
            for sy in (1, -1):
                a.append((sx_ * x) & mask); b.append((sy * y) & mask)
    for y in (1, 2, 3, top - 1, top, 12345, (1 << 56) + 1):
        for x in (0, 1, y - 1, y, y // 2, y // 2 + 1):
            if 0 <= x <= y:
                for sx_ in (1, -1):
                    for sy in (1, -1):
                        a.append((sx_ * x) & mask); b.append((sy * y) & mask)
    a.append(0); b.append(0)
    a = np.array(a, dtype=np.uint64); b = np.array(b, dtype=np.uint64)
    got, steps = gccpu.plain_op(OP["DIVB"], w, p, a, b)
    full, steps_full = gccpu.plain_op(OP["DIV"], w, p, a, b)
    assert np.array_equal(got, full)
    for x, y, g in zip(a[:400], b[:400], got[:400]):
        exp = oracle.div(_signed(int(x), w), _signed(int(y), w), p, w) & mask
        assert int(g) == exp, (p, hex(int(x)), hex(int(y)), hex(int(g)), hex(exp))
    assert steps == steps_full - 7 * (w - 1)


@pytest.mark.parametrize("w", [64, 32])
def test_idivc_matches_oracle(gccpu, oracle, w):
    """division by the public normalizer (a multiplication by a precomputed constant, Circ::divc) against tdiv: every
    divisor shape (1, powers of two, odd, just above / below a power of two, the largest ones) on random numerators, the
    extremes (0, +-1, INT_MAX, INT_MIN and its neighbours) and the numerators next to multiples of the divisor"""
    rng = np.random.default_rng(200 + w)
    top = (1 << (w - 1))
    mask = (1 << w) - 1
    big = [(1 << 31) - 1, (1 << 31) + 1, (1 << 32) - 1] if w == 64 else [(1 << 30) + 1, (1 << 31) - 1]
    steps_of = {}
    for c in [1, 2, 3, 5, 7, 20, 100, 127, 128, 129, 500, 641, 4096, 65535, 65537, 1000003] + big:
        a, _ = _operands(rng, w, 100)
        edge = [0, 1, mask, top - 1, top, top + 1, top - 2, 2, mask - 1]
        near = []
        for k in (1, 2, 3, (top - 1) // c, top // c, max(1, (top // c) - 1), int(rng.integers(1, 1 << 20))):
            for e in (-1, 0, 1):
                v = k * c + e
                if 0 <= v <= top:
                    near += [v, (-v) & mask]
        a = np.concatenate([a, np.array(edge + near, dtype=np.uint64)])
        got, steps = gccpu.plain_op(OP["IDIVC"], w, 7, a, None, c=c)
        steps_of[c] = steps
        for x, g in zip(a, got):
            exp = oracle.div(_signed(int(x), w), c, 0, w) & mask
            assert int(g) == exp, (w, c, hex(int(x)), hex(int(g)), hex(exp))
    assert steps_of[1] == 0
    # two conditional negates, one gate step per set bit of the multiplier beyond the first two, the final addition(s)
    adder = 1 + (6 if w == 64 else 5)
    assert max(steps_of.values()) <= 2 * adder + (w - 2) + 2 * 7
    assert steps_of[500] <= (64 if w == 64 else 40), steps_of


@pytest.mark.parametrize("w", [64, 32])
def test_sqrt_matches_oracle(gccpu, oracle, w):
    rng = np.random.default_rng(300 + w)
    for p in PRECS[w]:
        a, _ = _operands(rng, w, 200)
        sq = (np.arange(1, 40, dtype=np.uint64) ** 2)            # perfect squares and neighbours
        a = np.concatenate([a, sq, sq - np.uint64(1), sq + np.uint64(1)])
        got, steps = gccpu.plain_op(OP["SQRT"], w, p, a)
        for x, g in zip(a, got):
            exp = oracle.sqrt(_signed(x, w), p, w) & ((1 << w) - 1)
            assert int(g) == exp, (w, p, hex(int(x)), hex(int(g)), hex(exp))


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("w", [64, 32])
def test_mul_matches_oracle(gccpu, oracle, w, paired):
    rng = np.random.default_rng(400 + w)
    counts = set()
    for p in PRECS[w]:
        a, b = _operands(rng, w, 100)
        got, steps = gccpu.plain_op(OP["MUL"], w, p, a, b, paired=paired)
        counts.add(steps)
        for x, y, g in zip(a, b, got):
            exp = oracle.mul(_signed(x, w), _signed(y, w), p, w) & ((1 << w) - 1)
            assert int(g) == exp, (w, p, hex(int(x)), hex(int(y)))


REC = np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"),
                ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")])
OP_MAC, OP_MACK, OP_HDIFF = 1, 20, 21


def _mac_reference(oracle, a, b, p):
    """sum_k wrap_64((a_k b_k) >> p) mod 2^64 with the oracle's multiplication (src/fixed.oc:149-162)"""
    tot = 0
    for x, y in zip(a, b):
        tot += oracle.mul(_signed(x, 64), _signed(y, 64), p, 64)
    return tot & ((1 << 64) - 1)


@pytest.mark.parametrize("p", [56, 63, 33, 32, 31, 8, 1, 0, 48])
def test_karatsuba_mac_matches_oracle(gccpu, oracle, p):
    """OP_MACK (three 32 x 32 arrays per product, two products per wave) accumulates exactly what OP_MAC does:
    the sum of the oracle's truncated products, for every precision the command line accepts at w = 64, on edge
    operands (all-ones, sign boundaries, halves equal / zero) and random ones, even and odd chunk lengths."""
    rng = np.random.default_rng(900 + p)
    m = (1 << 64) - 1
    edge = [0, 1, m, m - 1, 1 << 63, (1 << 63) - 1, (1 << 63) + 1, 0xffffffff, 0x100000000, 0xffffffff00000000, 0x00000001ffffffff,
            0x8000000080000000, 0x7fffffff7fffffff, 0x5555555555555555, 0xaaaaaaaaaaaaaaaa, 0xfffffffe00000001, 0x123456789abcdef0]
    ea = [x for x in edge for _ in edge]; eb = [y for _ in edge for y in edge]
    ra = [int(v) for v in rng.integers(0, 1 << 63, size=400, dtype=np.uint64) * 2 + rng.integers(0, 2, size=400, dtype=np.uint64)]
    rb = [int(v) for v in rng.integers(0, 1 << 63, size=400, dtype=np.uint64) * 2 + rng.integers(0, 2, size=400, dtype=np.uint64)]
    # small-magnitude operands of either sign (what a fixed-point solve mostly sees)
    sa = [int(v) & m for v in rng.integers(-(1 << 40), 1 << 40, size=100)]
    sb = [int(v) & m for v in rng.integers(-(1 << 58), 1 << 58, size=100)]
    A = np.array(ea + ra + sa, dtype=np.uint64); Bv = np.array(eb + rb + sb, dtype=np.uint64)
    n = len(A)
    for cnt in (2, 7, 1, 20):
        groups = n // cnt
        # word file: 0 | a (n) | b (n) | hdiff a (n) | hdiff b (n) | results (2 per group, MACK) | results (2 per group, MAC)
        base_a, base_b = 1, 1 + n
        delta = 2 * n
        res_k, res_m = 1 + 4 * n, 1 + 4 * n + 2 * groups
        words = np.zeros(1 + 4 * n + 4 * groups, dtype=np.uint64)
        words[base_a:base_a + n] = A; words[base_b:base_b + n] = Bv
        recs = np.zeros(2 * n + 2 * groups, dtype=REC)
        step = 0
        sh, _ = gccpu.rec_cost(OP_HDIFF, 1, 64, p)
        for i in range(2 * n):
            recs[i] = (OP_HDIFF, 1, 1 + delta + i, 1 + i, 0, 0, 1, 1, step); step += sh
        sk, gk = gccpu.rec_cost(OP_MACK, cnt, 64, p)
        sm, gm = gccpu.rec_cost(OP_MAC, cnt, 64, p)
        for g in range(groups):
            recs[2 * n + g] = (OP_MACK, cnt, res_k + 2 * g, base_a + g * cnt, base_b + g * cnt, delta, 1, 1, step); step += sk
        for g in range(groups):
            recs[2 * n + groups + g] = (OP_MAC, cnt, res_m + 2 * g, base_a + g * cnt, base_b + g * cnt, 0, 1, 1, step); step += sm
        steps, gates = gccpu.plain_run(recs.view(np.uint8), len(recs), 64, p, words, np.zeros(1, dtype=np.uint64))
        assert steps == step
        for g in range(groups):
            exp = _mac_reference(oracle, A[g * cnt:(g + 1) * cnt], Bv[g * cnt:(g + 1) * cnt], p)
            got_k = (int(words[res_k + 2 * g]) + int(words[res_k + 2 * g + 1])) & m
            got_m = (int(words[res_m + 2 * g]) + int(words[res_m + 2 * g + 1])) & m
            assert got_m == exp, (p, cnt, g)
            assert got_k == exp, (p, cnt, g, [hex(int(v)) for v in A[g * cnt:(g + 1) * cnt]], [hex(int(v)) for v in Bv[g * cnt:(g + 1) * cnt]])
        if cnt % 2 == 0 and p == 56:
            assert sk < 0.88 * sm and gk < 0.92 * gm, (sk, sm, gk, gm)        # fewer gate steps AND fewer gates than the array
