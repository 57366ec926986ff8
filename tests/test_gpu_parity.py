"""GPU parity tests (MI355X): the HIP garble + evaluate path, called through
the C ABI, against the CPU semantic oracle on the same seeded inputs.
Bit-exact (integer work): no tolerance."""
import json
import os

import numpy as np
import pytest

from helpers import oracle_solve, split_shares, sx, synth_system

pytestmark = pytest.mark.gpu

FIPS_PT = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
FIPS_CT = bytes.fromhex("3925841d02dc09fbdc118597196a0b32")


def test_gpu_aes_known_answer(lgc, gccpu):
    assert lgc.device_count() >= 1
    pt = np.frombuffer(FIPS_PT, dtype=np.uint8)
    assert bytes(lgc.aes_encrypt(pt)[0]) == FIPS_CT
    rnd = np.random.default_rng(0).integers(0, 256, size=(4096 + 3, 16), dtype=np.uint8)
    assert np.array_equal(lgc.aes_encrypt(rnd), gccpu.aes(rnd))


def _solve(lgc, sysm, shares, seed=bytes(range(16)), profile=False):
    s = lgc.Solver(sysm, seed=seed)
    s.set_shares(shares)
    s.run(profile=profile)
    return s


CASES = [(64, 56, 5, 40), (64, 54, 8, 60), (32, 30, 6, 50), (32, 28, 4, 30), (64, 30, 3, 20)]


@pytest.mark.parametrize("w,p,d,n", CASES)
@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
@pytest.mark.parametrize("normalize", [0, 1])
def test_gpu_solver_matches_oracle(lgc, oracle, w, p, d, n, alg, normalize):
    rng = np.random.default_rng(w * 1000 + p * 10 + d + normalize)
    A, b = synth_system(oracle, rng, n, d, w, p)
    nsh = 3 if normalize else 2
    shares = split_shares(rng, A, b, nsh, w)
    iters, lam = 5, 0.001
    sysm = lgc.make_system(d, w, p, alg, iters, lam, nsh, normalize, reveal_inputs=1, trace=1)
    s = _solve(lgc, sysm, shares)
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=(alg == "cgd"))
    assert s.inputs().tolist() == np.concatenate([a, bb]).tolist()
    if alg == "cgd":
        beta, tr = exp
        assert s.trace().tolist() == tr.tolist()
    else:
        beta = exp
    assert s.beta().tolist() == beta.tolist()
    st = s.stats()
    assert st["and_gates"] > 0 and st["seconds_total"] > 0
    s.close()


def test_gpu_readme_known_answer(lgc, golden_dir):
    """README.md:85-87 of the reference, through garbling + evaluation on the GPU"""
    kat = json.load(open(os.path.join(golden_dir, "readme_kat.json")))
    d = 5
    A = np.array(kat["A"], dtype=np.uint64); b = np.array(kat["b"], dtype=np.uint64)
    shares = split_shares(np.random.default_rng(1), A, b, 3, 64)
    sysm = lgc.make_system(d, 64, 56, "cgd", 10, 0.001, 3, 1, 0, 1)
    s = _solve(lgc, sysm, shares)
    got = s.beta()
    assert got.tolist() == kat["beta_cgd"]
    assert ["%.15f" % (int(v) / 2.0 ** 56) for v in got] == kat["printed"]
    assert s.trace().tolist() == kat["trace"]
    # per-iteration marks (cgd.oc:190-194): equal gate increments, increasing device time
    g, t = s.iterations()
    total = s.stats()["and_gates"]
    assert len(g) == 10 and len(set(np.diff(g.astype(np.int64)).tolist())) == 1 and int(g[-1]) <= total
    assert np.all(np.diff(t) > 0) and t[-1] <= s.stats()["seconds_total"] + 1e-6


def test_gpu_matches_cpu_mirror_labels(lgc, gccpu, oracle):
    """different seeds give different garblings but the same decoded result;
    the same seed gives the same result as the CPU mirror of the protocol"""
    rng = np.random.default_rng(3)
    w, p, d, n = 64, 56, 3, 20
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, "cgd", 2, 0.0, 2, 0, 0, 0)
    outs = []
    for seed in (bytes(range(16)), bytes(range(1, 17))):
        s = _solve(lgc, sysm, shares, seed=seed)
        outs.append(s.beta().tolist())
        s.close()
    assert outs[0] == outs[1]
    prog = lgc.Program(sysm)
    dec, _, _ = gccpu.garble_eval(prog, shares, seed=bytes(range(16)))
    assert sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w).tolist() == outs[0]


def test_gpu_medium_dimension_chunked(lgc, oracle):
    rng = np.random.default_rng(5)
    w, p, d, n = 64, 56, 40, 300
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    for alg, iters in (("cgd", 3), ("cholesky", 0), ("ldlt", 0)):
        sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
        s = _solve(lgc, sysm, shares)
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, 0.0, 0)
        assert s.beta().tolist() == exp.tolist()
        s.close()


@pytest.mark.parametrize("w,p,alg", [(64, 56, "cholesky"), (64, 56, "cgd"), (32, 30, "ldlt")])
def test_split_and_four_wave_kernels_are_interchangeable(lgc, oracle, w, p, alg):
    """The latency-bound launches run column-split on 16 waves (garbler: critical-path garbling + table pass) or on
    4 waves per record (one-pass garbler, two-table AES; gc_split.h / gc_device.h): same gate numbering, tweaks and
    table rows, so any garbler kernel pairs with any evaluator kernel."""
    rng = np.random.default_rng(99)
    d, n = 7, 50
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, 4, 0.0, 2, 0, reveal_inputs=1, trace=1)
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, 4, 0.0, 0, trace=(alg == "cgd"))
    beta = exp[0] if alg == "cgd" else exp
    try:
        for g, e in [(1, 0), (0, 1), (0, 0), (1, 1)]:
            lgc.set_split_kernels(g, e)
            s = _solve(lgc, sysm, shares)
            assert s.beta().tolist() == beta.tolist(), (g, e)
            if alg == "cgd":
                assert s.trace().tolist() == exp[1].tolist(), (g, e)
            s.close()
    finally:
        lgc.set_split_kernels(1, 1)
