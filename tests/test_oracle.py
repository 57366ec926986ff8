"""CPU suite: the C semantic oracle against (i) the reference's only
known-answer (README.md:85-87) and (ii) golden vectors produced by the
independent pure-Python big-int twin (tests/golden/gen_golden.py)."""
import json
import os

import numpy as np

README_PRINTED = ["0.984331027786964", "0.792399824970372", "0.754117840176144",
                  "0.592849130685193", "0.057351715952213"]     # README.md:87
CANDIDATE_INTS = [70928525599209419, 57098424903440242, 54339917184171597,
                  42719281984652104, 4132626665463108]


def test_readme_known_answer(oracle, golden_dir):
    beta = oracle.linreg_file(os.path.join(golden_dir, "readme_example.in"),
                              56, -1, 64, 64, 2, 10, 0.001)
    printed = ["%.15f" % (int(b) / 2.0 ** 56) for b in beta]
    assert printed == README_PRINTED
    assert [int(b) for b in beta] == CANDIDATE_INTS


def test_readme_intermediates(oracle, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "readme_kat.json")))
    inp = oracle.read_input(os.path.join(golden_dir, "readme_example.in"))
    n, d = inp["n"], inp["d"]
    assert (n, d, inp["P"], inp["start"]) == (10, 5, 3, [0, 1, 2])
    Xq = oracle.quantize(inp["X"], 56, n, 64); yq = oracle.quantize(inp["y"], 56, n, 64)
    assert Xq.tolist() == kat["Xq"] and yq.tolist() == kat["yq"]
    A, b = oracle.aggregate(Xq, yq, n, d, 56, 64)
    assert [int(v) for v in A] == kat["A"] and [int(v) for v in b] == kat["b"]
    a, bb = oracle.circuit_input(oracle.sum_shares(A[None, :], 64), oracle.sum_shares(b[None, :], 64),
                                 d, 0.001, 56, 64)
    assert a.tolist() == kat["a"] and bb.tolist() == kat["bb"]
    beta, tr = oracle.cgd(a, bb, d, 56, 64, 10, trace=True)
    assert beta.tolist() == kat["beta_cgd"] and tr.tolist() == kat["trace"]
    assert oracle.cholesky(a, bb, d, 56, 64).tolist() == kat["beta_cholesky"]
    assert oracle.ldlt(a, bb, d, 56, 64).tolist() == kat["beta_ldlt"]


def test_ops_golden(oracle, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "ops.json")))
    assert len(cases) > 700
    for c in cases:
        w, p = c["w"], c["p"]
        if "ip" in c:
            assert oracle.inner_product(c["va"], c["vb"], p, w) == c["ip"], c
            continue
        a, b = c["a"], c["b"]
        assert oracle.add(a, b, w) == c["add"]
        assert oracle.sub(a, b, w) == c["sub"]
        assert oracle.abs(a, w) == c["abs"]
        assert oracle.cmp(a, b, w) == c["cmp"]
        assert oracle.mul(a, b, p, w) == c["mul"], c
        assert oracle.div(a, b, p, w) == c["div"], c
        assert oracle.sqrt(a, p, w) == c["sqrt"], c


def test_systems_golden(oracle, golden_dir):
    systems = json.load(open(os.path.join(golden_dir, "systems.json")))
    for s in systems:
        n, d, w, p = s["n"], s["d"], s["w"], s["p"]
        Xq = oracle.quantize(s["X"], p, n, w); yq = oracle.quantize(s["y"], p, n, w)
        assert Xq.tolist() == s["Xq"] and yq.tolist() == s["yq"]
        A, b = oracle.aggregate(Xq, yq, n, d, p, w)
        assert [int(v) for v in A] == s["A"] and [int(v) for v in b] == s["b"]
        a, bb = oracle.circuit_input(oracle.sum_shares(A[None, :], w), oracle.sum_shares(b[None, :], w),
                                     d, s["lam"], p, w)
        assert a.tolist() == s["a"] and bb.tolist() == s["bb"]
        assert oracle.cgd(a, bb, d, p, w, s["iters"]).tolist() == s["beta"]["cgd"]
        assert oracle.cholesky(a, bb, d, p, w).tolist() == s["beta"]["cholesky"]
        assert oracle.ldlt(a, bb, d, p, w).tolist() == s["beta"]["ldlt"]


def _rand_system(rng, n, d, p, w, oracle):
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    y = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
    return oracle.quantize(X, p, n, w), oracle.quantize(y, p, n, w)


def test_share_simulations_sum_to_aggregate(oracle):
    rng = np.random.default_rng(7)
    for (n, d, p, w, start) in ((12, 5, 56, 64, [0, 1, 2]), (9, 6, 30, 32, [0, 3]),
                                (7, 7, 56, 64, [0, 2, 4, 6]), (5, 4, 56, 64, [0])):
        Xq, yq = _rand_system(rng, n, d, p, w, oracle)
        A, b = oracle.aggregate(Xq, yq, n, d, p, w)
        m = (1 << w) - 1
        rnd = rng.integers(0, 2 ** 63, size=200000, dtype=np.uint64) * np.uint64(2) + \
            rng.integers(0, 2, size=200000, dtype=np.uint64)
        for fn in (oracle.ti_shares, oracle.ot_shares):
            sA, sb, used = fn(Xq, yq, n, d, p, w, start, rnd)
            assert [int(v) & m for v in sA.sum(axis=0)] == [int(v) for v in A]
            assert [int(v) & m for v in sb.sum(axis=0)] == [int(v) for v in b]
            if len(start) > 1:
                assert used > 0
                # shares individually look random: the first cross entry differs from the total
                assert not np.array_equal(sA[0], A)


def test_width_conversion_is_per_share(oracle):
    # phase1.c:609-638: each share is shifted on its own, so the recombined
    # value may differ from the shifted total by up to (#shares - 1) ulp.
    s = np.array([[0x7fffffffffffffff], [0x0000000004000001]], dtype=np.uint64)
    conv = np.stack([oracle.convert_shares(r, 56, 30, 64, 32) for r in s])
    total = (int(s[0, 0]) + int(s[1, 0])) & (2 ** 64 - 1)
    tot_conv = int(oracle.convert_shares(np.array([total], dtype=np.uint64), 56, 30, 64, 32)[0])
    got = (int(conv[0, 0]) + int(conv[1, 0])) & 0xffffffff
    assert got in (tot_conv, (tot_conv - 1) & 0xffffffff, (tot_conv + 1) & 0xffffffff)


def test_division_by_zero_definition(oracle):
    for w, p in ((64, 56), (32, 30)):
        assert oracle.div(5, 0, p, w) == -1 and oracle.div(-5, 0, p, w) == 1 and oracle.div(0, 0, p, w) == -1
