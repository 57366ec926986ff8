"""Edge cases of the domain on the GPU path vs the oracle: smallest dimensions, a single share,
extreme precisions, convergence to a zero residual (division by zero, unspecified in the reference
and defined identically in oracle and circuit), extreme operand values."""
import os

import numpy as np
import pytest

from helpers import oracle_solve, split_shares, synth_system

pytestmark = pytest.mark.gpu


def _check(lgc, oracle, A, b, d, w, p, alg, iters, nshares, normalize, lam, rng):
    shares = split_shares(rng, A, b, nshares, w) if nshares > 1 else np.concatenate([A, b])[None, :].astype(np.uint64)
    sysm = lgc.make_system(d, w, p, alg, iters, lam, nshares, normalize, 1, 1 if alg == "cgd" else 0)
    s = lgc.Solver(sysm, seed=bytes(range(7, 23)))
    s.set_shares(shares)
    s.run()
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=(alg == "cgd"))
    assert s.inputs().tolist() == np.concatenate([a, bb]).tolist()
    if alg == "cgd":
        assert s.trace().tolist() == exp[1].tolist()
        exp = exp[0]
    assert s.beta().tolist() == exp.tolist()
    s.close()


@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
def test_dimension_one_and_single_share(lgc, oracle, alg):
    rng = np.random.default_rng(0)
    A, b = synth_system(oracle, rng, 12, 1, 64, 56)
    _check(lgc, oracle, A, b, 1, 64, 56, alg, 3, 1, 1, 0.01, rng)
    A, b = synth_system(oracle, rng, 12, 2, 32, 30)
    _check(lgc, oracle, A, b, 2, 32, 30, alg, 3, 1, 0, 0.0, rng)


def test_cgd_past_convergence_divides_by_zero(lgc, oracle):
    """d=2 converges in two steps; after that the residual is exactly zero and every division has
    a zero divisor.  Oracle and circuit share one definition of that case."""
    rng = np.random.default_rng(1)
    for w, p in ((64, 56), (32, 30)):
        A, b = synth_system(oracle, rng, 30, 2, w, p)
        _check(lgc, oracle, A, b, 2, w, p, "cgd", 12, 2, 0, 0.0, rng)
    # an all-zero system: division by zero from the first step
    z = np.zeros(3, dtype=np.uint64), np.zeros(2, dtype=np.uint64)
    _check(lgc, oracle, z[0], z[1], 2, 64, 56, "cgd", 3, 2, 0, 0.0, rng)


@pytest.mark.parametrize("w,p", [(64, 1), (64, 60), (64, 63), (32, 1), (32, 31), (32, 29)])
def test_extreme_precisions_cgd(lgc, oracle, w, p):
    rng = np.random.default_rng(w + p)
    d = 3
    T = d * (d + 1) // 2
    m = (1 << w) - 1
    # arbitrary words (not a meaningful regression problem): pure arithmetic parity, wrap-around included
    A = rng.integers(0, 2 ** 63, size=T, dtype=np.uint64) & np.uint64(m)
    b = rng.integers(0, 2 ** 63, size=d, dtype=np.uint64) & np.uint64(m)
    _check(lgc, oracle, A, b, d, w, p, "cgd", 3, 2, 0, 0.0, rng)


@pytest.mark.parametrize("w,p", [(64, 63), (64, 62), (64, 61), (64, 60), (64, 2), (32, 31), (32, 30), (32, 3)])
def test_extreme_precisions_cholesky_ldlt(lgc, oracle, w, p):
    """square root with odd and even datapath widths (the 32-bit loop of fixed.oc:228-240 at odd
    32+p, the 64-bit non-restoring form at its widest single-word width (p = 60) and the two-word
    remainder for precision 61..63, all allowed by src/cmd/linreg.c:85-88), garbage operands included"""
    rng = np.random.default_rng(100 + w + p)
    d = 3
    T = d * (d + 1) // 2
    m = (1 << w) - 1
    A = rng.integers(0, 2 ** 63, size=T, dtype=np.uint64) & np.uint64(m)
    b = rng.integers(0, 2 ** 63, size=d, dtype=np.uint64) & np.uint64(m)
    for alg in ("cholesky", "ldlt"):
        _check(lgc, oracle, A, b, d, w, p, alg, 0, 2, 1, 0.5, rng)


def test_extreme_operand_values(lgc, oracle):
    rng = np.random.default_rng(9)
    for w, p in ((64, 56), (32, 30)):
        lo = 1 << (w - 1)
        specials = [0, 1, (1 << w) - 1, lo, lo - 1, lo + 1, 1 << (w - 2)]
        d = 3
        for rep in range(3):
            A = np.array([specials[(rep + k) % len(specials)] for k in range(6)], dtype=np.uint64)
            b = np.array([specials[(2 * rep + k + 3) % len(specials)] for k in range(3)], dtype=np.uint64)
            _check(lgc, oracle, A, b, d, w, p, "cgd", 2, 2, 0, 0.0, rng)


@pytest.mark.parametrize("w,p", [(64, 56), (32, 30)])
@pytest.mark.parametrize("d", [2, 3, 7, 16, 33])
def test_normalizer_on_extreme_sums(lgc, oracle, w, p, d):
    """division by the public normalizer d (linear.oc:52-65; Circ::divc multiplies by a precomputed constant) on the sums
    that stress it: INT_MIN and its neighbours, +-1, multiples of d and their neighbours, arbitrary words -- through the
    narrow kernels (d < 32) and the one-wave-per-record kernel (d = 33: 594 records)"""
    rng = np.random.default_rng(w + d)
    T = d * (d + 1) // 2
    m = (1 << w) - 1
    top = 1 << (w - 1)
    special = [top, top + 1, top - 1, m, 1, 0, 2, m - 1]
    for k in (1, 5, top // d, top // d - 1, int(rng.integers(1, 1 << 20))):
        for e in (-1, 0, 1):
            v = k * d + e
            if 0 <= v <= top:
                special += [v, (-v) & m]
    vals = np.array(special, dtype=np.uint64)
    A = rng.integers(0, 2 ** 63, size=T, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=T, dtype=np.uint64)
    b = rng.integers(0, 2 ** 63, size=d, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=d, dtype=np.uint64)
    A &= np.uint64(m); b &= np.uint64(m)
    off = [i * (i + 1) // 2 + j for i in range(d) for j in range(i)]          # the normalizer divides these (and b)
    for t, v in zip(off, vals):
        A[t] = v
    for i in range(d):
        b[i] = vals[(len(off) + i) % len(vals)]
    _check(lgc, oracle, A, b, d, w, p, "cgd", 1, 3, 1, 0.25, rng)


def test_parked_table_ring_is_reused_and_released(lgc, oracle):
    """a closed solver parks its table ring for the next one (gc_engine.hip: RingCache): a larger system after a smaller
    one, a smaller one after a larger one and a solve after lgc_release_cached_memory() all give the oracle's result"""
    rng = np.random.default_rng(41)
    for d in (6, 12, 4):
        A, b = synth_system(oracle, rng, 40, d, 64, 56)
        _check(lgc, oracle, A, b, d, 64, 56, "cholesky", 0, 2, 0, 0.0, rng)
    lgc.release_cached_memory()
    lgc.release_cached_memory()                      # nothing parked: a no-op
    A, b = synth_system(oracle, rng, 40, 5, 64, 56)
    _check(lgc, oracle, A, b, 5, 64, 56, "cgd", 3, 2, 0, 0.0, rng)


def test_rejects_unsupported_parameters(lgc):
    with pytest.raises(lgc.LgcError):
        lgc.Solver(lgc.make_system(3, width=64, precision=64))      # p < width (src/cmd/linreg.c:85-88)
    with pytest.raises(lgc.LgcError):
        lgc.Solver(lgc.make_system(0))
    s = lgc.Solver(lgc.make_system(2, algorithm="cgd", num_iterations=1))
    with pytest.raises(lgc.LgcError):
        s.run()                      # shares not set
    with pytest.raises(lgc.LgcError):
        s.beta()                     # not run


def test_gpu_randomised_parity_sweep():
    """40 random configurations (tests/tools/gpu_fuzz.py) against the oracle, bit for bit"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_fuzz.py"), "40", "99"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "40 cases, 0 mismatches" in r.stdout
