"""Parity at BASELINE.json's full sizes: the oracle is plain integer arithmetic and finishes
d = 500 in seconds, so these are exact comparisons, not just invariants."""
import numpy as np
import pytest

from helpers import oracle_solve, split_shares, synth_system

pytestmark = pytest.mark.gpu


def _run(lgc, oracle, n, d, w, p, alg, iters, nshares, normalize, lam, seed):
    rng = np.random.default_rng(seed)
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, nshares, w)
    sysm = lgc.make_system(d, w, p, alg, iters, lam, nshares, normalize, 0, 1 if alg == "cgd" else 0)
    s = lgc.Solver(sysm, seed=bytes(range(16)))
    s.set_shares(shares)
    s.run()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=(alg == "cgd"))
    if alg == "cgd":
        assert s.trace().tolist() == exp[1].tolist()
        exp = exp[0]
    assert s.beta().tolist() == exp.tolist()
    st = s.stats()
    s.close()
    return st


def test_config2_cholesky_d20(lgc, oracle):
    """BASELINE config 2: n=1000 d=20 cholesky 64-bit, two parties co-located"""
    _run(lgc, oracle, 1000, 20, 64, 56, "cholesky", 0, 2, 0, 0.0, 2)


def test_config3_cgd15_d100(lgc, oracle):
    """BASELINE config 3 (phase 2): d=100 cgd 15 iterations 64-bit, data-provider input path"""
    _run(lgc, oracle, 2000, 100, 64, 56, "cgd", 15, 2, 1, 1e-3, 3)


def test_headline_cgd15_d500_bit_exact(lgc, oracle):
    """the bench workload itself: d=500 CGD-15, 64-bit, every per-iteration reveal compared"""
    st = _run(lgc, oracle, 1500, 500, 64, 56, "cgd", 15, 2, 0, 0.0, 4)
    assert 2.5e10 < st["and_gates"] <= 2.9e10          # Karatsuba matrix-vector products (3.22e10 with the plain array)


def test_config4_cgd20_d500_32bit(lgc, oracle):
    """BASELINE config 4 (phase 2): d=500 cgd 20 iterations in 32-bit / precision 30, 5 providers"""
    _run(lgc, oracle, 1500, 500, 32, 30, "cgd", 20, 5, 1, 1e-3, 5)


def test_ldlt_d60(lgc, oracle):
    _run(lgc, oracle, 800, 60, 64, 56, "ldlt", 0, 2, 1, 1e-3, 6)


@pytest.mark.parametrize("d,w,p,iters", [(110, 64, 56, 2), (111, 64, 56, 2), (130, 64, 56, 2), (300, 64, 56, 2), (300, 32, 30, 3), (181, 32, 30, 2)])
def test_mid_size_cgd_around_the_karatsuba_threshold(lgc, oracle, d, w, p, iters):
    """round 4 made the records of the matrix-vector launches as short as the circuit allows (gc_program.h: kMvRecords64/32) while
    Karatsuba records are still used where d * d exceeds 12 288 -- d = 110 is the last plain system, 111 the first with
    Karatsuba pairs (one pair per record from there on), 300 the size where the old and the new record length differ most;
    32-bit: two-chunk records of one product per half at d = 181.  Every reveal of every iteration against the oracle."""
    _run(lgc, oracle, 3 * d, d, w, p, "cgd", iters, 2, 1, 1e-3, 40 + d)


@pytest.mark.parametrize("alg,d", [("cholesky", 150), ("ldlt", 190)])
def test_mid_size_factorisations_with_short_column_records(lgc, oracle, alg, d):
    """column steps of a factorisation with more than 4 096 products use Karatsuba records of one pair (kFactRecords); d = 150
    has such steps in the middle columns only, 190 nearly everywhere"""
    _run(lgc, oracle, 3 * d, d, 64, 56, alg, 0, 2, 1, 1e-3, 60 + d)
