"""Tolerance pins from material the reference itself holds (tests/golden/gen_reference_pins.py):

* experiments/matlab/phillipp4.data, adria1.data: A, b and the solution the reference's fixed-point
  study compares against;
* the `error` columns of experiments/results/phase2_{32,64}/*.out: what the reference's own 32- and
  64-bit CGD / Cholesky circuits achieved on the generate_tests.py distribution.

These are the only reference-held numbers that constrain Cholesky / LDL^T / sqrt and the 32-bit
solvers at all (exact vectors exist for CGD at 64 bits only: README.md:85-87).  They are float
tolerances, stated in each test; exact parity of the HIP path is against the oracle."""
import collections
import json
import os

import numpy as np
import pytest

from helpers import sx

HERE = os.path.dirname(os.path.abspath(__file__))


def _quant(oracle, v, p, w):
    return np.array([oracle.lib.orc_double_to_fixed(float(x), p, w) for x in np.ravel(v)], dtype=np.int64)


def _tri(A):
    d = A.shape[0]
    return np.array([A[i, j] for i in range(d) for j in range(i + 1)])


def _oracle_solve(oracle, a, b, d, p, w, alg, iters):
    if alg == "cgd":
        return oracle.cgd(a, b, d, p, w, iters)
    return oracle.cholesky(a, b, d, p, w) if alg == "cholesky" else oracle.ldlt(a, b, d, p, w)


def _gpu_solve(lgc, a, b, d, p, w, alg, iters):
    m = np.uint64((1 << w) - 1)
    tot = np.concatenate([a, b]).astype(np.uint64) & m
    mask = (np.arange(tot.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & m      # test_linear_system.c masks both shares
    with np.errstate(over="ignore"):
        shares = np.stack([mask, (tot - mask) & m])
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
    s = lgc.Solver(sysm, seed=bytes(range(16)))
    s.set_shares(shares)
    s.run()
    out = s.beta().copy()
    s.close()
    return out


# ------------------------------------------------------------------ MATLAB systems
MATLAB = json.load(open(os.path.join(HERE, "golden", "matlab_systems.json")))
# iterations as experiments/matlab/test7.m:4-11 sets them for each file
MATLAB_ITERS = {"phillipp4.data": 10, "adria1.data": 20}
# |beta - z|_inf bound.  z is printed with 12 significant digits and the systems have condition numbers
# of a few hundred, so z itself solves A z = b only to ~1e-8; CGD is additionally limited by its
# iteration count (the MATLAB study plots exactly this error, test7.m:47).
MATLAB_TOL = {("phillipp4.data", "cgd"): 2e-4, ("adria1.data", "cgd"): 2e-3}
DIRECT_TOL = 5e-7


def _matlab_case(oracle, sysd, p=56, w=64):
    A = np.array(sysd["A"]); b = np.array(sysd["b"]); z = np.array(sysd["z"])
    assert np.abs(A - A.T).max() < 1e-12
    return _quant(oracle, _tri(A), p, w), _quant(oracle, b, p, w), z


@pytest.mark.parametrize("sysd", MATLAB, ids=[s["name"] for s in MATLAB])
@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
def test_oracle_solves_matlab_systems(oracle, sysd, alg):
    a, b, z = _matlab_case(oracle, sysd)
    d = sysd["d"]
    beta = _oracle_solve(oracle, a, b, d, 56, 64, alg, MATLAB_ITERS[sysd["name"]]) / 2.0 ** 56
    tol = MATLAB_TOL.get((sysd["name"], alg), DIRECT_TOL)
    assert np.abs(beta - z).max() < tol, (alg, np.abs(beta - z).max())


@pytest.mark.gpu
@pytest.mark.parametrize("sysd", MATLAB, ids=[s["name"] for s in MATLAB])
@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
def test_gpu_solves_matlab_systems(lgc, oracle, sysd, alg):
    a, b, z = _matlab_case(oracle, sysd)
    d = sysd["d"]
    iters = MATLAB_ITERS[sysd["name"]]
    got = _gpu_solve(lgc, a, b, d, 56, 64, alg, iters)
    assert got.tolist() == _oracle_solve(oracle, a, b, d, 56, 64, alg, iters).tolist()      # exact vs the oracle
    tol = MATLAB_TOL.get((sysd["name"], alg), DIRECT_TOL)
    assert np.abs(got / 2.0 ** 56 - z).max() < tol


# ------------------------------------------------------------------ error magnitudes of the reference's runs
def _ref_errors():
    t = collections.defaultdict(list)
    for r in json.load(open(os.path.join(HERE, "golden", "reference_errors.json"))):
        t[(r["width"], r["alg"], r["d"])].append(r)
    return t


REF = _ref_errors()
PREC = {32: 30, 64: 56}      # experiments/test_phase2_aws.py:393


def _instance(seed, n, d, sigma=0.1):
    """generate_lin_system_from_regression_problem (experiments/generate_tests.py:134-146, 159-169)"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    beta = rng.random(d)
    y = X @ beta + sigma * rng.standard_normal(n)
    lam = sigma ** 2 / (n * np.linalg.norm(beta) ** 2)
    A = X.T @ X / (d * n) + np.eye(d) * lam
    b = X.T @ y / (d * n)
    x = np.linalg.solve(A, b)
    # the solvers read A and b from the text file write_system produced with Python 2's str(): 12
    # significant digits (generate_tests.py:15,20), while `solution` is the full-precision solve --
    # this rounding, not the fixed-point arithmetic, is the floor of the reference's 64-bit errors
    txt = np.vectorize(lambda v: float("%.12g" % v))
    return txt(A), txt(b), x


# our error must land in the band the reference's three instances per configuration span, widened by
# this factor either way (the published errors of one configuration agree within +-20 %)
BAND = 3.0
CASES = [(64, "cgd", 10), (64, "cgd", 20), (64, "cgd", 50), (64, "cholesky", 10), (64, "cholesky", 20), (64, "cholesky", 50),
         (32, "cgd", 10), (32, "cgd", 20), (32, "cgd", 50), (32, "cholesky", 10), (32, "cholesky", 20), (32, "cholesky", 50)]


@pytest.mark.parametrize("w,alg,d", CASES)
def test_oracle_error_matches_reference_runs(oracle, w, alg, d):
    refs = [r["error"] for r in REF[(w, alg, d)]]
    assert len(refs) >= 2
    p = PREC[w]
    errs = []
    for seed in (0, 1):
        A, b, x = _instance(seed * 100 + d, 100000, d)
        a, bq = _quant(oracle, _tri(A), p, w), _quant(oracle, b, p, w)
        beta = _oracle_solve(oracle, a, bq, d, p, w, alg, 20) / 2.0 ** p
        errs.append(float(np.linalg.norm(beta - x)))
    assert min(refs) / BAND <= min(errs) and max(errs) <= max(refs) * BAND, (errs, refs)


def test_oracle_cgd_error_curve_matches_reference_runs(oracle):
    """per-iteration error of CGD (iter_i / error_i rows): same decay as the reference's 64-bit d=20 runs"""
    ref = np.array([r["iter_errors"] for r in REF[(64, "cgd", 20)]])
    A, b, x = _instance(7, 100000, 20)
    a, bq = _quant(oracle, _tri(A), 56, 64), _quant(oracle, b, 56, 64)
    _, tr = oracle.cgd(a, bq, 20, 56, 64, 20, trace=True)
    ours = np.linalg.norm(tr[:, :20] / 2.0 ** 56 - x, axis=1)
    lo, hi = ref.min(axis=0), ref.max(axis=0)
    # early iterations depend on the instance (|beta|); the decay rate and the floor do not
    assert np.all(ours[5:] <= hi[5:] * 30) and np.all(ours[5:] >= lo[5:] / 30), (ours, lo, hi)
    assert ours[-1] <= hi[-1] * BAND


@pytest.mark.gpu
@pytest.mark.parametrize("w,alg,d", [(64, "cgd", 20), (64, "cholesky", 20), (32, "cgd", 20), (32, "cholesky", 20), (64, "ldlt", 20)])
def test_gpu_error_matches_reference_runs(lgc, oracle, w, alg, d):
    p = PREC[w]
    A, b, x = _instance(d, 100000, d)
    a, bq = _quant(oracle, _tri(A), p, w), _quant(oracle, b, p, w)
    got = _gpu_solve(lgc, a, bq, d, p, w, alg, 20)
    assert sx(got.astype(np.uint64), w).tolist() == _oracle_solve(oracle, a, bq, d, p, w, alg, 20).tolist()
    refs = [r["error"] for r in REF[(w, "cholesky" if alg == "ldlt" else alg, d)]]
    err = float(np.linalg.norm(sx(got.astype(np.uint64), w) / 2.0 ** p - x))
    assert min(refs) / (BAND * (3 if alg == "ldlt" else 1)) <= err <= max(refs) * BAND * (3 if alg == "ldlt" else 1), (err, refs)


def test_reference_gate_count_polynomials_match_every_published_file(golden_dir):
    """lgc_reference_gate_count reproduces the gate_count column of every
    experiments/results/phase2_{32,64}/*.out of the current circuits (all six d values of the 64-bit runs, both
    algorithms) and, for cgd, the cumulative count after EVERY iteration -- the figure bench.py's
    ref_equiv_gates_per_s and the ref_gate_count column of python/results.py are built on."""
    import json
    import linreg_gc
    recs = json.load(open(os.path.join(golden_dir, "reference_errors.json")))
    seen = set()
    for r in recs:
        # phase2_64/ also holds *_32_* files of an older 32-bit implementation (SURVEY.md 6.1 note): not the current fixed.oc
        if ("_%d_" % r["width"]) not in r["file"] or not r["file"].startswith("phase2_%d/" % r["width"]):
            continue
        total = linreg_gc.reference_gate_count(r["alg"], r["width"], r["d"], r["iters"] if r["alg"] == "cgd" else 0)
        assert total == r["gate_count"], r["file"]
        if r["alg"] == "cgd":
            for it, g in enumerate(r["iter_gates"], start=1):
                assert linreg_gc.reference_gate_count("cgd", r["width"], r["d"], it) == g, (r["file"], it)
        seen.add((r["alg"], r["width"], r["d"]))
    assert {d for a, w, d in seen if a == "cgd" and w == 64} == {10, 20, 50, 100, 200, 500}
    assert {d for a, w, d in seen if a == "cholesky" and w == 64} >= {20, 100, 500}
    assert linreg_gc.reference_gate_count("ldlt", 64, 10) is None
    assert linreg_gc.reference_gate_count("cgd", 64, 500, 15) == 74149847180
