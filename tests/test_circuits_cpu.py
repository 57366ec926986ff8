"""Op-level checks of the wave-shaped circuits (gc_circuits.h, plaintext backend) against the
semantic oracle: random and edge operands for every word operation, every width / precision the
command line accepts (src/cmd/linreg.c:85-88), in both step orders the GPU kernels use."""
import numpy as np
import pytest

OP = dict(MUL=7, ADD=9, SUB=10, ABS=11, DIV=13, SQRT=14, IDIVC=15)


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


@pytest.mark.parametrize("w", [64, 32])
def test_idivc_matches_oracle(gccpu, oracle, w):
    rng = np.random.default_rng(200 + w)
    for c in (1, 2, 3, 5, 20, 100, 500, 4096):
        a, _ = _operands(rng, w, 100)
        got, _ = gccpu.plain_op(OP["IDIVC"], w, 7, a, None, c=c)
        for x, g in zip(a, got):
            exp = oracle.div(_signed(x, w), c, 0, w) & ((1 << w) - 1)
            assert int(g) == exp, (w, c, hex(int(x)))


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
