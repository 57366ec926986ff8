"""CPU suite for the host logic: the lowered program of the product library
(gc_program.h via the C ABI) is executed record by record on the CPU checker
and compared with the semantic oracle.  No GPU needed."""
import numpy as np
import pytest

from helpers import oracle_solve, split_shares, sx, synth_system

FIPS_KEY_PT = bytes.fromhex("3243f6a8885a308d313198a2e0370734")   # FIPS-197 Appendix B
FIPS_KEY_CT = bytes.fromhex("3925841d02dc09fbdc118597196a0b32")


def test_aes_known_answer(gccpu):
    pt = np.frombuffer(FIPS_KEY_PT, dtype=np.uint8)
    assert bytes(gccpu.aes(pt)[0]) == FIPS_KEY_CT            # AES-NI
    assert bytes(gccpu.aes(pt, ttable=True)[0]) == FIPS_KEY_CT  # the T-table algorithm the GPU runs
    rnd = np.random.default_rng(0).integers(0, 256, size=(257, 16), dtype=np.uint8)
    assert np.array_equal(gccpu.aes(rnd), gccpu.aes(rnd, ttable=True))


REC = np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"),
                ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")])


def _plain(lgc, gccpu, sysm, shares):
    prog = lgc.Program(sysm)
    info = prog.info
    words = np.zeros(info.n_words, dtype=np.uint64)
    m = (1 << sysm.width) - 1
    words[info.in_base:info.in_base + shares.size] = shares.ravel() & np.uint64(m)
    dec = np.zeros(info.n_reveal + 1, dtype=np.uint64)
    steps, gates = gccpu.plain_run(prog.records(), info.n_records, sysm.width, sysm.precision, words, dec)
    assert steps == info.total_steps and gates == info.total_gates
    return prog, dec


def test_cpu_checker_hash_matches_openssl(gccpu):
    """oracle/gc_cpu.cpp is compiled from the product's hash header, so its agreement with the GPU is self-consistency; here its
    gate hash is checked against the definition written over OpenSSL's AES (tests/helpers.py: openssl_gate_hash) -- independent
    code, independent AES"""
    from helpers import openssl_gate_hash
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, size=(257, 16), dtype=np.uint8)
    t = rng.integers(0, 2 ** 63, size=len(x), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=len(x), dtype=np.uint64)
    x[0] = 0; t[0] = 0; x[1] = 255; t[1] = np.uint64(2 ** 64 - 1)
    assert np.array_equal(gccpu.gate_hash(x, t), openssl_gate_hash(x, t))


@pytest.mark.parametrize("p", [1, 30, 56, 63])
def test_cgd_on_arbitrary_words_keeps_the_bounded_divider_exact(lgc, gccpu, oracle, p):
    """CGD at 64 bit divides g by max|g| with the short divider (OP_DIVB: the dividend is never larger than the divisor).
    That rests on the program, not on the data: arbitrary 64-bit words as inputs -- products and sums that wrap, INT_MIN
    among the gradients, all-equal and all-zero vectors -- through the plaintext run of the whole program against the oracle,
    every per-iteration reveal included; and the program holds 2 x d short dividers per iteration + d, and no full
    divider on those operands"""
    w, iters = 64, 3
    rng = np.random.default_rng(900 + p)
    top = np.uint64(1 << 63)
    for case in range(12):
        d = int(rng.integers(1, 7))
        T = d * (d + 1) // 2
        A = rng.integers(0, 2 ** 63, size=T, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=T, dtype=np.uint64)
        b = rng.integers(0, 2 ** 63, size=d, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=d, dtype=np.uint64)
        if case == 0: b[:] = top                                  # g = -b = INT_MIN in every component
        if case == 1: b[0] = top; b[1:] = np.uint64(1)
        if case == 2: b[:] = 0; A[:] = 0                          # 0 / 0 from the first division on
        if case == 3: b[:] = np.uint64((1 << 64) - 1)             # g = +1 everywhere: quotient exactly 2^p
        if case == 4: A[:] = top
        shares = split_shares(rng, A, b, 2, w)
        sysm = lgc.make_system(d, w, p, "cgd", iters, 0.0, 2, 0, 1, 1)
        prog, dec = _plain(lgc, gccpu, sysm, shares)
        recs = np.frombuffer(prog.records().tobytes(), dtype=REC)
        assert (recs["op"] == 23).sum() == d * (iters + 1) and (recs["op"] == 13).sum() == 2 * iters      # OP_DIVB; OP_DIV: eta, gamma
        (beta, tr), a, bb = oracle_solve(oracle, A, b, d, w, p, "cgd", iters, 0.0, 0, trace=True)
        info = prog.info
        assert dec[info.rv_trace:info.rv_trace + iters * (d + 4)].astype(np.int64).tolist() == np.asarray(tr).ravel().tolist(), (case, d)
        assert sx(dec[info.rv_beta:info.rv_beta + d], w).tolist() == beta.tolist(), (case, d)


def test_dimension_check_program(lgc, gccpu):
    """the in-circuit comparison of the two parties' dimensions (src/linear.oc:109-114: revealOblivBool(feedOblivInt(d, 1) ==
    feedOblivInt(d, 2))) as a program of its own: one OP_EQ over two 32-bit input words -- 31 AND gates, what a comparison of
    two 32-bit words costs the reference -- and one reveal; on the plaintext machine and through CPU garbling + evaluation"""
    sysm = lgc.make_system(1, 32, 0, "dimcheck", 0, 0.0, 2, 0, 0, 0)
    prog = lgc.Program(sysm)
    info = prog.info
    assert (info.total_gates, info.total_steps, info.n_reveal) == (31, 5, 1)
    for a, b in ((500, 500), (500, 501), (0, 0), (0, 1 << 31), (0xffffffff, 0xffffffff), (0x80000000, 0), (12345, 12345 ^ (1 << 17))):
        shares = np.array([[a, 0], [b, 0]], dtype=np.uint64)
        _, dec = _plain(lgc, gccpu, sysm, shares)
        assert int(dec[info.rv_beta]) == (1 if a == b else 0), (a, b)
        got, gates, _ = gccpu.garble_eval(prog, shares, seed=bytes(range(16)))
        assert gates == 31 and int(got[info.rv_beta]) == (1 if a == b else 0), (a, b)
    with pytest.raises(lgc.LgcError):
        lgc.Program(lgc.make_system(2, 32, 0, "dimcheck", 0, 0.0, 2, 0, 0, 0))      # a program of its own: d = 1 only


CASES = [(64, 56, 5, 40), (64, 54, 7, 60), (32, 30, 6, 50), (32, 28, 4, 30), (64, 30, 3, 20)]


@pytest.mark.parametrize("w,p,d,n", CASES)
@pytest.mark.parametrize("alg", ["cgd", "cholesky", "ldlt"])
@pytest.mark.parametrize("normalize", [0, 1])
def test_program_plain_matches_oracle(lgc, gccpu, oracle, w, p, d, n, alg, normalize):
    rng = np.random.default_rng(w * 1000 + p * 10 + d + normalize)
    A, b = synth_system(oracle, rng, n, d, w, p)
    nsh = 3 if normalize else 2
    shares = split_shares(rng, A, b, nsh, w)
    iters = 6
    lam = 0.001
    sysm = lgc.make_system(d, w, p, alg, iters, lam, nsh, normalize, reveal_inputs=1, trace=1)
    prog, dec = _plain(lgc, gccpu, sysm, shares)
    info = prog.info
    exp, a, bb = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=(alg == "cgd"))
    T = d * (d + 1) // 2
    got_ab = sx(dec[info.rv_inputs:info.rv_inputs + T + d], w)
    assert got_ab.tolist() == np.concatenate([a, bb]).tolist()
    got = sx(dec[info.rv_beta:info.rv_beta + d], w)
    if alg == "cgd":
        beta, tr = exp
        got_tr = sx(dec[info.rv_trace:info.rv_trace + iters * (d + 4)], w).reshape(iters, d + 4)
        assert got_tr.tolist() == tr.tolist()
    else:
        beta = exp
    assert got.tolist() == beta.tolist()


@pytest.mark.parametrize("alg,d,iters", [("cgd", 132, 2), ("cholesky", 184, 0), ("ldlt", 184, 0)])
def test_karatsuba_lowering_plain_matches_oracle(lgc, gccpu, oracle, alg, d, iters):
    """the sizes from which the lowering uses Karatsuba records (two products per record: cgd from d = 130, the
    factorisations from d = 182): OP_HDIFF shadow words, OP_MACK records with even lengths and the zero-word pairing,
    through the plaintext run of the whole program, against the oracle -- and against the plain-array program, which
    must give the same integers with more gates"""
    rng = np.random.default_rng(d)
    w, p = 64, 56
    A, b = synth_system(oracle, rng, 2 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
    prog, dec = _plain(lgc, gccpu, sysm, shares)
    recs = np.frombuffer(prog.records().tobytes(), dtype=REC)
    # OP_MACK records, and the half-difference words of their operands: OP_HDIFF records (cgd) or -- since round 5 -- formed
    # by the division / multiplication record that makes a matrix entry final (cnt == 2: factorisations)
    hd = (recs["op"] == 21).sum() + ((recs["cnt"] == 2) & ((recs["op"] == 13) | (recs["op"] == 7))).sum()
    assert (recs["op"] == 20).sum() > 100 and hd >= d
    if alg != "cgd":
        assert (recs["op"] == 17).sum() == d * (d - 1) // 2          # OP_COPY: the input's mirror only -- no copy launch per column any more, the divider mirrors its quotient
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, 0.0, 0)
    got = sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w)
    assert got.tolist() == exp.tolist()
    try:
        lgc.set_karatsuba(False)
        plain, dec0 = _plain(lgc, gccpu, sysm, shares)
    finally:
        lgc.set_karatsuba(True)
    assert sx(dec0[plain.info.rv_beta:plain.info.rv_beta + d], w).tolist() == exp.tolist()
    assert plain.info.total_gates > prog.info.total_gates and plain.info.total_steps > prog.info.total_steps
    assert not np.isin(np.frombuffer(plain.records().tobytes(), dtype=REC)["op"], (20, 21)).any()        # OP_MACK, OP_HDIFF


@pytest.mark.parametrize("w,p,alg,d", [(32, 30, "cholesky", 250), (32, 30, "ldlt", 200), (64, 56, "cholesky", 150)])
def test_mid_size_factorisations_plain_match_oracle(lgc, gccpu, oracle, w, p, alg, d):
    """mid-size factorisations through the plaintext run (the small cases above never leave the first chunk size): the
    partial-sum scratch of a dots() call is sized for ANY batch -- at d = 250, 32 bit, the round-2 sizing let the records
    of the middle columns run past the end of the word file once the table cap grew (found on the GPU at d = 500) -- and
    lgc_program_build refuses a program whose records leave the word file"""
    rng = np.random.default_rng(d + w)
    A, b = synth_system(oracle, rng, 3 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    sysm = lgc.make_system(d, w, p, alg, 0, 1e-3, 2, 1, 0, 0)
    prog, dec = _plain(lgc, gccpu, sysm, shares)
    recs = np.frombuffer(prog.records().tobytes(), dtype=REC)
    assert int(recs["dst"].max()) < prog.info.n_words
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, 0, 1e-3, 1)
    assert sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w).tolist() == exp.tolist()


@pytest.mark.parametrize("d,nl", [(100, 64), (60, 32)])
def test_merged_sweep_at_config5_size_plain_matches_oracle(lgc, gccpu, oracle, d, nl):
    """BASELINE config 5's shape (d = 100, 64 lambdas; two CGD iterations here) through the plaintext run: the base
    program of a sweep is built for the merged launch -- few long Karatsuba records per circuit, the chunk bounded by the
    table cap, record-major replication, equal pieces -- a shape no small sweep reaches (an undersized partial-sum
    scratch in exactly this corner passed every small test and failed config 5 on the GPU)"""
    import sweep
    rng = np.random.default_rng(d)
    w, p, it = 64, 56, 2
    A, b = synth_system(oracle, rng, 3 * d, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    lams = sweep.c5_lambdas(64)[:nl]
    sysm = lgc.make_system(d, w, p, "cgd", it, 0.0, 2, 1, 0, 0)
    prog = lgc.Program(sysm, lambdas=lams)
    info = prog.info
    words = np.zeros(info.n_words, dtype=np.uint64)
    words[info.in_base:info.in_base + shares.size] = shares.ravel()
    dec = np.zeros(info.n_reveal + 1, dtype=np.uint64)
    steps, gates = gccpu.plain_run(prog.records(), info.n_records, w, p, words, dec)
    assert steps == info.total_steps and gates == info.total_gates
    recs = np.frombuffer(prog.records().tobytes(), dtype=REC)
    assert (recs["op"] == 20).sum() > 1000                              # Karatsuba records in the merged matrix-vector products
    for t_, lam in enumerate(lams):
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cgd", it, lam, 1)
        got = sx(dec[info.rv_beta + t_ * info.reveal_stride:info.rv_beta + t_ * info.reveal_stride + d], w)
        assert got.tolist() == exp.tolist(), (t_, lam)


def test_program_readme_example(lgc, gccpu, oracle, golden_dir):
    """the reference's only known answer, through the lowered circuit (plaintext run)"""
    import json, os
    kat = json.load(open(os.path.join(golden_dir, "readme_kat.json")))
    d = 5
    A = np.array(kat["A"], dtype=np.uint64); b = np.array(kat["b"], dtype=np.uint64)
    shares = split_shares(np.random.default_rng(1), A, b, 3, 64)
    sysm = lgc.make_system(d, 64, 56, "cgd", 10, 0.001, 3, 1, 0, 1)
    prog, dec = _plain(lgc, gccpu, sysm, shares)
    got = sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], 64)
    assert got.tolist() == kat["beta_cgd"]
    assert ["%.15f" % (int(v) / 2.0 ** 56) for v in got] == kat["printed"]


def test_program_medium_dimension(lgc, gccpu, oracle):
    """chunked dot products (several MAC records per row) and a deeper max tree"""
    rng = np.random.default_rng(5)
    w, p, d, n = 64, 56, 40, 300
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    for alg, iters in (("cgd", 3), ("cholesky", 0), ("ldlt", 0)):
        sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
        prog, dec = _plain(lgc, gccpu, sysm, shares)
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, 0.0, 0)
        assert sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w).tolist() == exp.tolist()


@pytest.mark.parametrize("w,p", [(64, 56), (32, 30)])
def test_cpu_garble_eval_matches_oracle(lgc, gccpu, oracle, w, p):
    """the half-gates protocol itself (CPU mirror of the GPU kernels)"""
    rng = np.random.default_rng(11 + w)
    d, n = 3, 25
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    for alg, iters in (("cgd", 2), ("cholesky", 0)):
        sysm = lgc.make_system(d, w, p, alg, iters, 0.001, 2, 1, 0, 0)
        prog = lgc.Program(sysm)
        dec, gates, _ = gccpu.garble_eval(prog, shares, seed=bytes(range(16)))
        assert gates == prog.info.total_gates
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, alg, iters, 0.001, 1)
        assert sx(dec[prog.info.rv_beta:prog.info.rv_beta + d], w).tolist() == exp.tolist()


@pytest.mark.parametrize("w,p,alg", [(64, 56, "cgd"), (32, 28, "cholesky"), (64, 54, "ldlt")])
def test_sweep_program_matches_oracle_per_lambda(lgc, gccpu, oracle, w, p, alg):
    """per-lambda sweep: several circuits replicated into one program (merged launches), each
    bit-exact against the oracle run with its own lambda"""
    rng = np.random.default_rng(77 + w)
    d, n, nsh, iters = 4, 30, 3, 4
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, nsh, w)
    lams = [0.0, 1e-6, 0.001, 0.37, 1.0]
    sysm = lgc.make_system(d, w, p, alg, iters, 123.0, nsh, 1, 0, 0)        # sys.lambda is ignored
    base = lgc.Program(lgc.make_system(d, w, p, alg, iters, lams[0], nsh, 1, 0, 0))
    prog = lgc.Program(sysm, lambdas=lams)
    info = prog.info
    # shared prefix: the constant zero, the input words and the share sums -- divided by the public normalizer in place,
    # off the diagonal and in b -- exist once (lambda enters on the diagonal, linear.oc:52-57); everything else is per
    # circuit: two prefix launches, the sums and the normalizer
    T = d * (d + 1) // 2
    se = info.shared_end
    assert se == base.info.shared_end == 1 + (nsh + 1) * (T + d) and info.prefix_launches == base.info.prefix_launches == 2
    recs = np.frombuffer(base.records().tobytes(), dtype=np.dtype([("op", "<u4"), ("cnt", "<u4"), ("dst", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("sa", "<i4"), ("sb", "<i4"), ("step0", "<u8")]))
    l1 = base.launches()[1]
    norm = recs[l1["first_rec"]:l1["first_rec"] + l1["nrec"]]
    assert (norm["op"] == 15).all() and len(norm) == T and (norm["dst"] == norm["a"]).all() and (norm["dst"] < se).all()   # OP_IDIVC
    assert not (recs[l1["first_rec"] + l1["nrec"]:]["op"] == 15).any()
    assert info.replicas == len(lams) and 0 < info.word_stride <= base.info.n_words - se    # (a merged circuit needs fewer, longer records)
    assert info.reveal_stride == base.info.n_reveal and info.n_reveal == len(lams) * base.info.n_reveal
    assert info.n_words == se + len(lams) * (base.info.n_words - se)
    pre_gates = sum(l["gates"] for l in base.launches()[:base.info.prefix_launches])
    assert info.total_gates == pre_gates + len(lams) * (base.info.total_gates - pre_gates)
    assert info.n_launches == base.info.n_launches                          # launches are merged, not appended
    words = np.zeros(info.n_words, dtype=np.uint64)
    m = np.uint64((1 << w) - 1)
    words[info.in_base:info.in_base + shares.size] = shares.ravel() & m      # ONE set of input words
    dec = np.zeros(info.n_reveal + 1, dtype=np.uint64)
    steps, gates = gccpu.plain_run(prog.records(), info.n_records, w, p, words, dec)
    assert steps == info.total_steps and gates == info.total_gates
    for t, lam in enumerate(lams):
        exp = oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, 1)[0]
        lo = info.rv_beta + t * info.reveal_stride
        assert sx(dec[lo:lo + d], w).tolist() == exp.tolist(), (t, lam)
    # blocks of a sharded sweep: same records, gate steps offset so that no two circuits of the whole sweep
    # share a gate id (the ranks share the prefix and with it the garbler's offset R)
    per = base.info.total_steps - base.info.prefix_steps
    stride = 1 << 36                                    # gc_program.h kSweepCircuitStride: the canonical gate-step stride of a circuit
    seen = []
    for first, cnt in ((0, 2), (2, 3)):
        blk = lgc.Program(sysm, lambdas=lams[first:first + cnt], first=first)
        L = blk.launches()
        npre = blk.info.prefix_launches
        assert [(l["step0"], l["steps"]) for l in L[:npre]] == [(l["step0"], l["steps"]) for l in prog.launches()[:npre]]
        lo = min(l["step0"] for l in L[npre:] if l["steps"]); hi = max(l["step0"] + l["steps"] for l in L[npre:])
        blk_per = (blk.info.total_steps - blk.info.prefix_steps) // cnt
        assert lo == base.info.prefix_steps + first * stride and hi == lo + cnt * blk_per and blk_per <= stride
        seen.append((lo, hi))
        words = np.zeros(blk.info.n_words, dtype=np.uint64)
        words[blk.info.in_base:blk.info.in_base + shares.size] = shares.ravel() & m
        dec = np.zeros(blk.info.n_reveal + 1, dtype=np.uint64)
        recs = np.frombuffer(blk.records().tobytes(), dtype=REC)
        # plain_run checks step0 against a running counter from 0: renumber the copy for the check
        recs = recs.copy()
        shift = recs["step0"] >= base.info.prefix_steps
        recs["step0"][shift] -= np.uint64(first * stride)
        gccpu.plain_run(recs.view(np.uint8), blk.info.n_records, w, p, words, dec)
        for t in range(cnt):
            exp = oracle_solve(oracle, A, b, d, w, p, alg, iters, lams[first + t], 1)[0]
            lo_r = blk.info.rv_beta + t * blk.info.reveal_stride
            assert sx(dec[lo_r:lo_r + d], w).tolist() == exp.tolist(), (first, t)
    assert seen[0][1] <= seen[1][0]
    with pytest.raises(RuntimeError):                                       # lambda only enters the DP input path
        lgc.Program(lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0), lambdas=lams)


@pytest.mark.parametrize("alg,d,w,p,iters,nl,world", [("cholesky", 200, 64, 56, 0, 5, 2), ("cgd", 100, 32, 28, 15, 27, 2),
                                                       ("cgd", 100, 64, 56, 15, 64, 7), ("cgd", 12, 64, 56, 3, 7, 3)])
def test_blocks_of_an_uneven_partition_have_disjoint_gate_steps(lgc, alg, d, w, p, iters, nl, world):
    """the blocks of one sharded sweep share the garbler's R, so no two of them may use a gate step twice -- also when
    the partition is uneven and the per-circuit step count therefore differs between blocks (the lowering sizes the
    dot-product records by the number of merged circuits): every block lies inside the canonical stride range of its
    circuits (round 3 laid block k at first * ITS OWN per-circuit count and the first two cases overlapped)"""
    from sweep import partition
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 1, 0, 0)
    lams = [10.0 ** (-6.0 + 6.0 * k / (nl - 1)) for k in range(nl)]
    stride, ranges, pers, prefix = 1 << 36, [], set(), None
    for r in range(world):
        lo, hi = partition(nl, world, r)
        blk = lgc.Program(sysm, lambdas=lams[lo:hi], first=lo)
        L = blk.launches()
        npre = blk.info.prefix_launches
        pre = [(l["step0"], l["steps"]) for l in L[:npre]]
        assert prefix is None or pre == prefix                              # one prefix, garbled once
        prefix = pre
        s0 = min(l["step0"] for l in L[npre:] if l["steps"]); s1 = max(l["step0"] + l["steps"] for l in L[npre:])
        assert s0 == blk.info.prefix_steps + lo * stride and s1 <= blk.info.prefix_steps + hi * stride
        ranges.append((s0, s1))
        pers.add((blk.info.total_steps - blk.info.prefix_steps) // (hi - lo))
        recs = np.frombuffer(blk.records().tobytes(), dtype=REC)
        body = recs[recs["step0"] >= blk.info.prefix_steps]
        assert body["step0"].min() >= s0 and body["step0"].max() <= s1      # (records without gates, e.g. reveals, may sit at the end)
    ranges.sort()
    assert all(a[1] <= b[0] for a, b in zip(ranges, ranges[1:])), ranges
    assert ranges[0][0] >= sum(s for _, s in prefix)
    if world == 2 and d >= 100:
        assert len(pers) == 2              # the case that matters: block sizes differ AND so do the per-circuit step counts


def test_library_exports_and_fails_loudly_without_gpu(lgc):
    import ctypes, re, os
    root = os.path.join(os.path.dirname(__file__), "..")
    L = lgc.lib()
    per_header = {}
    for h in ("linreg_gc.h", "linreg_gc_sweep.h", "linreg_gc_debug.h"):
        hdr = open(os.path.join(root, "include", h)).read()
        names = set(re.findall(r"^[a-z][^\n(]*?\b(lgc_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
        assert len(names) >= 10, h
        for nme in names:
            assert hasattr(L, nme), (h, nme)          # every declared entry point is exported by the library
        per_header[h] = names
    # the drop-in surface stays small, and INTEGRATION.md names the reference call behind every one of its entry points
    surface = per_header["linreg_gc.h"]
    assert len(surface) <= 70, len(surface)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert not [n for n in sorted(surface) if n not in doc]
    assert not (surface & per_header["linreg_gc_sweep.h"]) and not (surface & per_header["linreg_gc_debug.h"])
    if lgc.device_count() == 0:
        with pytest.raises(lgc.LgcError) as e:
            lgc.Solver(lgc.make_system(3))
        assert e.value.code == -2     # LGC_ENODEVICE: no CPU fallback
        with pytest.raises(lgc.LgcError) as e:
            lgc.devices_preflight([0])
        assert e.value.code == -2
    with pytest.raises(lgc.LgcError):
        lgc.Program(lgc.make_system(3, width=48))
    with pytest.raises(lgc.LgcError):
        lgc.Program(lgc.make_system(3, width=32, precision=32))


@pytest.mark.parametrize("alg,d,iters", [("cgd", 30, 3), ("cholesky", 14, 0), ("ldlt", 9, 0)])
def test_table_ring_plan_never_overwrites_live_tables(lgc, alg, d, iters):
    """host logic of the co-located solver's garbled-table ring: with the garbler as far ahead of the
    evaluator as its waits allow, no range is overwritten before its launch has been evaluated"""
    prog = lgc.Program(lgc.make_system(d, 64, 56, alg, iters, 0.0, 2, 0, 0, 0))
    L = prog.launches()
    steps = np.array([l["steps"] for l in L], dtype=np.int64)
    length = (steps * 2048 + 4095) // 4096 * 4096
    biggest = int(length.max())
    for ring in (0, biggest, biggest + 4096, 3 * biggest // 2, 64 * biggest):
        rb, off, wait = prog.ring_plan(ring)
        off = off.astype(np.int64)
        assert rb >= biggest and (ring == 0 or rb == max(ring, biggest))
        assert np.all(off + length <= rb) and np.all(wait < np.arange(len(L)))
        for i in range(len(L)):
            if length[i] == 0:
                continue
            lo = int(wait[i]) + 1                      # launches lo..i-1 may still be waiting for their evaluator
            live = np.arange(lo, i)
            live = live[length[live] > 0]
            clash = (off[live] < off[i] + length[i]) & (off[i] < off[live] + length[live])
            assert not clash.any(), (ring, i, live[clash][:4])
        if ring == 0:                                   # the default ring lets the garbler run ahead
            ahead = np.arange(len(L)) - wait - 1
            assert ahead.max() >= 2


def test_lowering_under_address_sanitizer(tmp_path):
    """the program builder and the record executor compiled with ASan + UBSan (CPU build: the GPU pool has no sanitizer
    runs): every configuration's word file is a heap block of exactly n_words entries, so one word past the builder's
    allocation is a report (tests/tools/asan_program.cpp; includes the sizes at which round 3's scratch overflow showed)"""
    import os, shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(__file__), "tools", "asan_program.cpp")
    exe = str(tmp_path / "asan_program")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", src, "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr + cc.stdout and "cannot find" in cc.stderr + cc.stdout:
        pytest.skip("sanitizer runtime not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, "full"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.rstrip().endswith("all ok"), (run.stdout[-1500:], run.stderr[-3000:])


def test_big_mac_launches_use_records_of_one_pair_and_small_systems_keep_their_shape(lgc):
    """round 4 (gc_program.h: kMvRecords64 / kMvRecords32 / kFactRecords): the matrix-vector product of d = 500 CGD is ONE launch
    of records of one Karatsuba pair each (records of one length retire in lock step and starve the other chain's small
    launches -- DESIGN.md 2.3 (ii)); Karatsuba is used where d * d > 8 192 (12 288 until the end of round 5: a Karatsuba launch of
    a few rounds now picks its waves per workgroup, which is what made it pay at d = 100), so d = 90 has no OP_MACK record
    and d = 91 has; non-MAC launches are still cut at 2^24 steps while MAC launches may reach 2^25"""
    OP_MACK, OP_MAC = 20, 1
    def shape(d, w, p, iters, alg="cgd", nshares=2, normalize=0):
        prog = lgc.Program(lgc.make_system(d, w, p, alg, iters, 0.0, nshares, normalize, 0, 0))
        recs = np.frombuffer(prog.records().tobytes(), dtype=REC)
        return prog.launches(), recs
    L, recs = shape(500, 64, 56, 2)
    mv = [l for l in L if recs["op"][l["first_rec"]] == OP_MACK]
    assert len(mv) == 2                                              # one launch per matrix-vector product
    for l in mv:
        r = recs[l["first_rec"]:l["first_rec"] + l["nrec"]]
        assert l["nrec"] == 125000 and set(r["cnt"].tolist()) == {2} and l["steps"] == 27500000 and l["steps"] <= 1 << 25
    assert max(l["steps"] for l in L if recs["op"][l["first_rec"]] not in (OP_MACK, OP_MAC)) <= 1 << 24
    L, recs = shape(90, 64, 56, 2)
    assert not (recs["op"] == OP_MACK).any()
    L, recs = shape(91, 64, 56, 2)
    mk = recs[recs["op"] == OP_MACK]
    assert len(mk) > 0 and set(mk["cnt"].tolist()) <= {4, 3, 2, 1}  # (an odd row length leaves one lone product per row; at most two pairs per record)
    L, recs = shape(250, 64, 56, 0, alg="cholesky", normalize=1)
    mk = recs[recs["op"] == OP_MACK]
    assert len(mk) > 0 and mk["cnt"].max() <= 4                    # (round 3: up to 22 products per record)
