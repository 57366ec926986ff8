"""Gate-hash option 1 (lgc_set_gate_hash: the Chaskey-12 permutation in the place of the fixed-key AES) on the GPU:
the hash itself against the CPU checker, and the same parity cases as the default hash -- the circuits, the gate
numbering and every revealed integer are the same, so the oracle comparisons are bit-exact as everywhere else."""
import numpy as np
import pytest

from helpers import oracle_solve, split_shares, sx, synth_system
from test_gpu_roles import _ot_setup

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chaskey(lgc, gccpu):
    lgc.set_gate_hash("chaskey12")
    old = gccpu.set_gate_hash(1)
    yield
    gccpu.set_gate_hash(old)
    lgc.set_gate_hash("aes128")


def test_gate_hash_on_device_matches_cpu_checker(lgc, gccpu):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(4096 + 5, 16), dtype=np.uint8)
    t = rng.integers(0, 2 ** 63, size=len(x), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=len(x), dtype=np.uint64)
    x[0] = 0; t[0] = 0
    for kind in (0, 1):
        assert np.array_equal(lgc.gate_hash_eval(kind, x, t), gccpu.gate_hash(kind, x, t)), kind
    assert not np.array_equal(lgc.gate_hash_eval(0, x[:8], t[:8]), lgc.gate_hash_eval(1, x[:8], t[:8]))


CASES = [(64, 56, 5, 40), (32, 30, 6, 50), (64, 30, 3, 20)]


@pytest.mark.parametrize("w,p,d,n", CASES)
@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
@pytest.mark.parametrize("normalize", [0, 1])
def test_gpu_solver_with_chaskey_hash_matches_oracle(lgc, oracle, chaskey, w, p, d, n, alg, normalize):
    rng = np.random.default_rng(w * 1000 + p * 10 + d + normalize)
    A, b = synth_system(oracle, rng, n, d, w, p)
    nsh = 3 if normalize else 2
    shares = split_shares(rng, A, b, nsh, w)
    iters, lam = 5, 0.001
    sysm = lgc.make_system(d, w, p, alg, iters, lam, nsh, normalize, reveal_inputs=1, trace=1)
    s = lgc.Solver(sysm, seed=bytes(range(16)))
    s.set_shares(shares)
    s.run()
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=(alg == "cgd"))
    assert s.inputs().tolist() == np.concatenate([a, bb]).tolist()
    if alg == "cgd":
        assert s.trace().tolist() == exp[1].tolist()
        exp = exp[0]
    assert s.beta().tolist() == exp.tolist()
    s.close()


def test_gpu_labels_with_chaskey_hash_match_cpu_mirror(lgc, gccpu, oracle, chaskey):
    """same seed, same hash: the GPU's decoded result equals the CPU mirror's, and differs in its garbling from the AES run
    only (the integers are the same)"""
    rng = np.random.default_rng(3)
    w, p, d, n = 64, 56, 3, 20
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, "cgd", 2, 0.0, 2, 0, 0, 0)
    s = lgc.Solver(sysm, seed=bytes(range(16))); s.set_shares(shares); s.run()
    out = s.beta().tolist()
    s.close()
    prog = lgc.Program(sysm)
    assert prog.info.gate_hash == 1
    dec, _, _ = gccpu.garble_eval(prog, shares, seed=bytes(range(16)))
    assert sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w).tolist() == out


@pytest.mark.parametrize("w,p,alg,d,iters", [(64, 56, "cgd", 132, 2), (32, 30, "cgd", 100, 2), (64, 56, "cholesky", 60, 0), (64, 56, "ldlt", 40, 0)])
def test_chaskey_hash_mid_sizes_reach_the_mac_and_wide_kernels(lgc, oracle, chaskey, w, p, alg, d, iters):
    """d = 132 CGD: Karatsuba MAC kernel (OP_MACK launches); d = 100 at 32 bits: the two-per-wave MAC; the factorisations:
    MAC launches of every size, wide launches of dividers, 4-wave launches -- all over the table-free hash"""
    rng = np.random.default_rng(d)
    A, b = synth_system(oracle, rng, 4 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 1 if alg == "cgd" else 0)
    s = lgc.Solver(sysm, seed=bytes(range(16))); s.set_shares(shares); s.run()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, 0.0, 0, trace=(alg == "cgd"))
    if alg == "cgd":
        assert s.trace().tolist() == exp[1].tolist()
        exp = exp[0]
    assert s.beta().tolist() == exp.tolist()
    s.close()


def test_split_roles_with_chaskey_hash(lgc, oracle, chaskey):
    """CSP and Evaluator as separate objects, inputs through the OT extension (whose own hash stays AES), tables handed
    over launch by launch; then a sweep of three lambdas as one program"""
    w, p, alg = 64, 56, "cgd"
    rng = np.random.default_rng(77)
    d, n, P, iters, lam = 4, 30, 3, 3, 0.001
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, P, w)
    sysm = lgc.make_system(d, w, p, alg, iters, lam, P, 1, reveal_inputs=1, trace=1)
    G = lgc.Party(sysm, lgc.GARBLER, seed=bytes(range(16)), max_launch_table_bytes=1 << 20)
    E = lgc.Party(sysm, lgc.EVALUATOR, max_launch_table_bytes=1 << 20)
    for k in range(P):
        seeds0, seeds1, delta, seeds_s = _ot_setup(rng)
        S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
        m0, m1 = G.input_pairs(k)
        choice = lgc.share_choice_bits(shares[k], w)
        labels = R.labels_finish(S.labels(m0, m1, R.labels_start(choice)))
        E.set_input_labels(k, labels)
        S.close(); R.close()
    for i in range(G.num_launches):
        E.evaluate(i, G.garble(i))
    beta, trace, inputs = E.finish(G.decode_bits())
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, 1, trace=True)
    assert inputs.tolist() == np.concatenate([a, bb]).tolist() and trace.tolist() == exp[1].tolist() and beta.tolist() == exp[0].tolist()
    G.close(); E.close()
    lams = [0.001, 0.01, 0.1]
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, P, 1, 0, 0)
    s = lgc.Solver(sysm, seed=bytes(range(16)), lambdas=lams); s.set_shares(shares); s.run()
    got = s.beta()
    for t, l in enumerate(lams):
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, l, 1)
        assert got[t].tolist() == exp.tolist()
    s.close()


def test_a_role_with_the_other_hash_does_not_decode(lgc, oracle):
    """the two roles must agree on the hash: an evaluator built for AES against a garbler built for Chaskey-12 produces
    labels that decode to something else (no crash, no silent match)"""
    rng = np.random.default_rng(9)
    w, p, d = 64, 56, 3
    A, b = synth_system(oracle, rng, 20, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, "cholesky", 0, 0.0, 2, 0, 0, 0)
    lgc.set_gate_hash("chaskey12")
    try:
        G = lgc.Party(sysm, lgc.GARBLER, seed=bytes(range(16)))
        E1 = lgc.Party(sysm, lgc.EVALUATOR)
    finally:
        lgc.set_gate_hash("aes128")
    E = lgc.Party(sysm, lgc.EVALUATOR)
    # what the host binaries compare before the first table moves (host/tables.c: programs_agree)
    assert G.program_fingerprint() == E1.program_fingerprint() != E.program_fingerprint()
    E1.close()
    E2 = lgc.Party(lgc.make_system(d, w, p, "cholesky", 0, 0.5, 2, 1, 0, 0), lgc.EVALUATOR)     # another lambda
    E3 = lgc.Party(lgc.make_system(d, w, p, "cholesky", 0, 0.25, 2, 1, 0, 0), lgc.EVALUATOR)
    assert E2.program_fingerprint() != E3.program_fingerprint()
    E2.close(); E3.close()
    ok = True
    try:
        for k in range(2):
            m0, m1 = G.input_pairs(k)
            choice = lgc.share_choice_bits(shares[k], w)
            E.set_input_labels(k, np.where(choice[:, None] == 1, m1, m0))
        for i in range(G.num_launches):
            E.evaluate(i, G.garble(i))
        beta, _, _ = E.finish(G.decode_bits())
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cholesky", 0, 0.0, 0)
        ok = beta.tolist() == exp.tolist()
    finally:
        G.close(); E.close()
    assert not ok


def test_headline_workload_with_chaskey_hash_bit_exact(lgc, oracle, chaskey):
    """d = 500 CGD-15, 64-bit, every per-iteration reveal compared: the bench's `alt_hash` leg"""
    rng = np.random.default_rng(4)
    n, d, w, p = 1500, 500, 64, 56
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, "cgd", 15, 0.0, 2, 0, 0, 1)
    s = lgc.Solver(sysm, seed=bytes(range(16))); s.set_shares(shares); s.run()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cgd", 15, 0.0, 0, trace=True)
    assert s.trace().tolist() == exp[1].tolist() and s.beta().tolist() == exp[0].tolist()
    s.close()
