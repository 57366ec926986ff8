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


def test_gate_hash_on_device_matches_openssl(lgc):
    """the T-table AES of the kernels (LDS tables built at compile time, v_perm addressing) inside the gate hash, against the
    hash written from its definition over OpenSSL's AES (tests/helpers.py): nothing shared with the product or its CPU mirror.
    And one AND gate garbled and evaluated by hand with that hash obeys the half-gates equations the kernels implement
    (ZRE15): what the evaluator computes from (a, b, TG, TE) is c0 ^ (va & vb) R."""
    from helpers import openssl_gate_hash
    rng = np.random.default_rng(9)
    x = rng.integers(0, 256, size=(2048 + 3, 16), dtype=np.uint8)
    t = rng.integers(0, 2 ** 63, size=len(x), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=len(x), dtype=np.uint64)
    x[0] = 0; t[0] = 0
    assert np.array_equal(lgc.gate_hash_eval(x, t), openssl_gate_hash(x, t))
    # half-gates by hand, hashes from the DEVICE: garbler side
    R = rng.integers(0, 256, size=16, dtype=np.uint8); R[0] |= 1
    a0 = rng.integers(0, 256, size=16, dtype=np.uint8); b0 = rng.integers(0, 256, size=16, dtype=np.uint8)
    gid = 123456789
    H = lambda lab, tw: lgc.gate_hash_eval(lab[None, :], np.array([tw], dtype=np.uint64))[0]
    pa, pb = int(a0[0] & 1), int(b0[0] & 1)
    h0, h1, h2, h3 = H(a0, 2 * gid), H(a0 ^ R, 2 * gid), H(b0, 2 * gid + 1), H(b0 ^ R, 2 * gid + 1)
    TG = h0 ^ h1 ^ (R if pb else 0 * R)
    WG = h0 ^ (TG if pa else 0 * TG)
    TE = h2 ^ h3 ^ a0
    WE = h2 ^ ((TE ^ a0) if pb else 0 * TE)
    c0 = WG ^ WE
    for va in (0, 1):
        for vb in (0, 1):
            a = a0 ^ (R if va else 0 * R); b = b0 ^ (R if vb else 0 * R)
            sa, sb = int(a[0] & 1), int(b[0] & 1)
            got = openssl_gate_hash(a[None, :], [2 * gid])[0] ^ (TG if sa else 0 * TG) ^ openssl_gate_hash(b[None, :], [2 * gid + 1])[0] ^ ((TE ^ a) if sb else 0 * TE)
            assert np.array_equal(got, c0 ^ (R if (va & vb) else 0 * R)), (va, vb)


def test_garbled_tables_of_a_circuit_match_half_gates_written_over_openssl(lgc):
    """Label-level check that shares nothing with the product: the 31-gate dimension comparison (LGC_ALG_DIMCHECK, OP_EQ) is
    garbled on the GPU, and every ciphertext row of its five gate steps is recomputed here from the garbler's input label pairs
    with half-gates (ZRE15) written over OpenSSL's AES -- the hash, the tweak numbering gid = 64 step + lane, the (TG, TE) row
    layout, the free-XOR glue and lane moves of the circuit, and the chaining of output labels from step to step.
    (oracle/gc_cpu.cpp cannot give that: it is compiled from the product's headers.)"""
    from helpers import openssl_gate_hash
    sysm = lgc.make_system(1, 32, 0, "dimcheck", 0, 0.0, 2, 0, 0, 0)
    prog = lgc.Program(sysm)
    G = lgc.Party(sysm, 1, seed=bytes(range(32, 48)))
    A0, A1 = G.input_pairs(0)
    B0, B1 = G.input_pairs(1)
    R = A0[0] ^ A1[0]
    assert R[0] & 1 and all(np.array_equal(A0[i] ^ A1[i], R) and np.array_equal(B0[i] ^ B1[i], R) for i in range(64))
    Ls = prog.launches()
    k = [i for i, L in enumerate(Ls) if L["steps"]][0]
    assert Ls[k]["steps"] == 5 and G.table_bytes(k) == 5 * 2048
    tab = G.garble(k).reshape(5, 2, 64, 16)                      # step, (TG | TE), lane, label
    H = lambda lab, tw: openssl_gate_hash(lab[None, :], [tw])[0]
    Z = np.zeros(16, dtype=np.uint8)
    nz = [A0[l] ^ B0[l] for l in range(32)] + [Z] * 32           # zero-labels of a ^ b (word 0 of either share), lanes 32.. are 0
    step = int(Ls[k]["step0"])
    for s_i, dist in enumerate((16, 8, 4, 2, 1)):
        sh = [nz[l + dist] if l + dist < 64 else Z for l in range(64)]
        t = [Z] * 64
        for l in range(64):
            if l >= dist:
                assert not tab[s_i, :, l].any()                  # inactive lanes: zero rows
                continue
            a0, b0, gid = nz[l], sh[l], 64 * step + l
            pa, pb = int(a0[0] & 1), int(b0[0] & 1)
            h0, h1, h2, h3 = H(a0, 2 * gid), H(a0 ^ R, 2 * gid), H(b0, 2 * gid + 1), H(b0 ^ R, 2 * gid + 1)
            TG = h0 ^ h1 ^ (R if pb else Z)
            TE = h2 ^ h3 ^ a0
            assert np.array_equal(tab[s_i, 0, l], TG) and np.array_equal(tab[s_i, 1, l], TE), (s_i, l)
            t[l] = h0 ^ (TG if pa else Z) ^ h2 ^ ((TE ^ a0) if pb else Z)
        nz = [(nz[l] ^ sh[l] ^ t[l]) if l < dist else Z for l in range(64)]
        step += 1
    # and the evaluator's side of the same tables: equal inputs decode to 1, different ones to 0
    for da, db in ((77, 77), (77, 78)):
        E = lgc.Party(sysm, 2)
        E.set_input_labels(0, G.encode_inputs(0, [da, 0]))
        E.set_input_labels(1, G.encode_inputs(1, [db, 0]))
        for i in range(G.num_launches):
            E.evaluate(i, tab.reshape(-1) if i == k else G.garble(i))
        beta, _, _ = E.finish(G.decode_bits())
        assert int(beta[0]) == (1 if da == db else 0)
        E.close()
    G.close()
