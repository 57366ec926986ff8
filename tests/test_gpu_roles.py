"""CSP and Evaluator as separate objects with data-provider input sharing through the GPU OT
extension -- the whole phase-2 flow of src/cmd/linreg.c:145-199 + src/input.c, bytes carried by
the test instead of sockets -- against the oracle."""
import numpy as np
import pytest

from helpers import oracle_solve, split_shares, synth_system

pytestmark = pytest.mark.gpu


def _ot_setup(rng):
    seeds0 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
    seeds1 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
    delta = rng.integers(0, 256, size=16, dtype=np.uint8)
    dbits = np.unpackbits(delta, bitorder="little")
    return seeds0, seeds1, delta, np.where(dbits[:, None] == 1, seeds1, seeds0)


@pytest.mark.parametrize("w,p,alg", [(64, 56, "cgd"), (32, 30, "cholesky"), (64, 56, "ldlt")])
def test_split_roles_with_ot_inputs(lgc, oracle, w, p, alg):
    rng = np.random.default_rng(w + len(alg))
    d, n, P, iters, lam = 4, 30, 3, 3, 0.001
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, P, w)
    sysm = lgc.make_system(d, w, p, alg, iters, lam, P, 1, reveal_inputs=1, trace=1)
    small = 1 << 20                                   # 1 MiB of tables per launch: many launches
    G = lgc.Party(sysm, lgc.GARBLER, seed=bytes(range(16)), max_launch_table_bytes=small)
    E = lgc.Party(sysm, lgc.EVALUATOR, max_launch_table_bytes=small)
    assert G.num_launches == E.num_launches and G.num_launches > 10
    for k in range(P):                                # one IKNP session per data provider (input.c:59-69)
        seeds0, seeds1, delta, seeds_s = _ot_setup(rng)
        S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
        m0, m1 = G.input_pairs(k)                                       # CSP: yaoKeyNewPair per bit
        choice = lgc.share_choice_bits(shares[k], w)                    # DP: its share bits
        u = R.labels_start(choice)
        e = S.labels(m0, m1, u)
        labels = R.labels_finish(e)                                     # DP obtains one label per bit ...
        assert np.array_equal(labels, np.where(choice[:, None] == 1, m1, m0))
        E.set_input_labels(k, labels)                                   # ... and forwards them to the Evaluator
        S.close(); R.close()
    total = 0
    for i in range(G.num_launches):
        t = G.garble(i)
        total += t.size
        E.evaluate(i, t)
    beta, trace, inputs = E.finish(G.decode_bits())
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, 1, trace=(alg == "cgd"))
    assert inputs.tolist() == np.concatenate([a, bb]).tolist()
    if alg == "cgd":
        assert trace.tolist() == exp[1].tolist()
        exp = exp[0]
    assert beta.tolist() == exp.tolist()
    assert total == sum(G.table_bytes(i) for i in range(G.num_launches))
    with pytest.raises(lgc.LgcError):
        G.finish(G.decode_bits())                     # results are revealed to party 2 only
    G.close(); E.close()


def test_two_party_path_garbler_encodes_own_input(lgc, oracle):
    """test_linear_system path (linear.oc:96-135): party 1 feeds its own words directly,
    party 2's words go through the OT"""
    rng = np.random.default_rng(2)
    w, p, d, n = 64, 56, 3, 20
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, "cgd", 2, 0.0, 2, 0, 0, 0)
    G = lgc.Party(sysm, lgc.GARBLER, seed=bytes(range(3, 19))); E = lgc.Party(sysm, lgc.EVALUATOR)
    E.set_input_labels(0, G.encode_inputs(0, shares[0]))
    seeds0, seeds1, delta, seeds_s = _ot_setup(rng)
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    m0, m1 = G.input_pairs(1)
    u = R.labels_start(lgc.share_choice_bits(shares[1], w))
    E.set_input_labels(1, R.labels_finish(S.labels(m0, m1, u)))
    for i in range(G.num_launches):
        E.evaluate(i, G.garble(i))
    beta, _, _ = E.finish(G.decode_bits())
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cgd", 2, 0.0, 0)
    assert beta.tolist() == exp.tolist()
