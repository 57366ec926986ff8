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


def test_table_ring_holds_ciphertexts_only(lgc):
    """the hipIpc table ring is mapped by the Evaluator process: at no moment may it hold the garbler's zero-labels
    (one zero-label next to the active label of the same wire gives R).  Critical-path launches (dividers, square
    roots, max trees: record kernel + table pass) keep their stash in the garbler's private buffer, so the record
    kernel leaves the slot untouched -- all-zero on first use -- and the table pass fills it with exactly the bytes
    the socket path sends (src/input.c:94-108: label pairs never leave the CSP)."""
    sysm = lgc.make_system(5, 64, 56, "cholesky", 0, 0.001, 2, 1)
    seed = bytes(range(7, 23))
    G = lgc.Party(sysm, lgc.GARBLER, seed=seed); ref = lgc.Party(sysm, lgc.GARBLER, seed=seed)
    nslots = 64
    G.ring_create(nslots)
    crit_launches, steps_checked = 0, 0
    for k in range(G.num_launches):
        nb = G.table_bytes(k)
        before = G.test_ring_read(k, nb)
        if k < nslots:
            assert not before.any()                              # zero-filled at creation, never used
        crit = G.test_garble_ring_stage(k, 1)                    # record kernel only
        after = G.test_ring_read(k, nb)
        expect = ref.garble(k)                                   # the same launch on the socket path, same seed
        if crit:
            crit_launches += 1
            steps_checked += nb // 2048
            assert np.array_equal(after, before)                 # nothing reached the shared slot yet
            G.test_garble_ring_stage(k, 2)                       # table pass
            after = G.test_ring_read(k, nb)
        if crit or k < nslots:
            assert np.array_equal(after, expect)
        else:                                                    # reused slot: lanes without a gate keep older ciphertexts
            live = expect != 0
            assert np.array_equal(after[live], expect[live])
    assert crit_launches >= 10 and steps_checked > 5000           # sqrt / div launches of the 5 columns
    G.close(); ref.close()


@pytest.mark.parametrize("streams", [2, 1])
def test_asynchronous_ring_garbling_gives_the_tables_of_the_socket_path(lgc, streams):
    """(streams = 1, lgc_party_garble_ring_streams: the table passes stay on the record kernels' stream, one stash.)
    lgc_party_garble_ring_begin / _wait (what host/tables.c drives from two threads): two dozen launches of a Cholesky solve at a time are
    enqueued without a single wait in between -- record kernels back to back on one stream, the table passes of the
    critical-path launches on another, the zero-label stash alternating between two buffers -- and every launch, once waited
    for, holds exactly the bytes the socket path sends for the same seed.  Several batches: slots, events and both stashes are reused."""
    sysm = lgc.make_system(10, 64, 56, "cholesky", 0, 0.001, 2, 1)
    seed = bytes(range(9, 25))
    G = lgc.Party(sysm, lgc.GARBLER, seed=seed); ref = lgc.Party(sysm, lgc.GARBLER, seed=seed)
    nslots, batch = 64, 24
    G.ring_create(nslots)
    n = G.num_launches
    assert n > 2 * batch
    with pytest.raises(lgc.LgcError):
        G.garble_ring_wait(0)                                    # nothing was begun
    with pytest.raises(lgc.LgcError):
        G.garble_ring_streams(3)
    if streams == 1:
        G.garble_ring_streams(1)
    checked = 0
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        for k in range(lo, hi):
            G.garble_ring_begin(k)
        with pytest.raises(lgc.LgcError):
            G.garble_ring_wait(hi)                               # not begun yet
        for k in range(lo, hi):
            G.garble_ring_wait(k)
            nb = G.table_bytes(k)
            got, expect = G.test_ring_read(k, nb), ref.garble(k)
            if k < nslots:
                assert np.array_equal(got, expect), k
            else:                                                # reused slot: lanes without a gate keep older ciphertexts
                live = expect != 0
                assert np.array_equal(got[live], expect[live]), k
            checked += nb
    assert checked > 10 << 20
    with pytest.raises(lgc.LgcError):
        G.garble_ring_streams(2)                                 # launches have been begun
    G.close(); ref.close()


def test_ti_ring_messages_are_truncated_to_32_bits(lgc, oracle):
    """--ti_ring --width_phase1=32: the masked vectors b + x and a - y are written to device memory the peer provider
    maps.  X is stored sign-extended to 64 bits, so the kernels must truncate to the protocol width: bits 32..63 of a
    word would otherwise carry the sign of the private datum plus a carry (the socket path masks on the host)."""
    import ctypes as C
    rng = np.random.default_rng(11)
    n, d, p, w = 257, 3, 30, 32
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    Xq = oracle.quantize(X, p, n, w).reshape(n, d)
    assert (Xq < 0).any()
    L = lgc.lib()
    L.lgc_p1_set_device_io.argtypes = [C.c_void_p, C.c_int]; L.lgc_p1_set_device_io.restype = C.c_int
    L.lgc_p1_ti_a_batch.argtypes = [C.c_void_p] * 2 + [C.c_size_t] + [C.c_void_p] * 5; L.lgc_p1_ti_a_batch.restype = C.c_int
    for f, a in (("lgc_dev_alloc", [C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]), ("lgc_dev_upload", [C.c_void_p, C.c_void_p, C.c_size_t]),
                 ("lgc_dev_download", [C.c_void_p, C.c_void_p, C.c_size_t])):
        getattr(L, f).argtypes = a; getattr(L, f).restype = C.c_int
    L.lgc_dev_free.argtypes = [C.c_void_p]; L.lgc_dev_free.restype = None
    ph = lgc.Phase1(Xq, None, w, p)
    npairs = 3
    V = rng.integers(0, 1 << 32, size=(npairs, n), dtype=np.uint64)
    inn = rng.integers(0, 1 << 32, size=(npairs, n), dtype=np.uint64)
    cols = np.array([0, 2, 1], dtype=np.uint32)
    host_pos = ph.mask(cols, V, +1); host_neg = ph.mask(cols, V, -1)          # host path: masked on the way out
    nbytes = npairs * n * 8
    dV, dIn, dOut = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr in (dV, dIn, dOut):
        assert L.lgc_dev_alloc(0, nbytes, C.byref(ptr), None) == 0
    assert L.lgc_dev_upload(dV, V.ctypes.data_as(C.c_void_p), nbytes) == 0
    assert L.lgc_dev_upload(dIn, inn.ctypes.data_as(C.c_void_p), nbytes) == 0
    assert L.lgc_p1_set_device_io(ph._h, 1) == 0
    out = np.zeros((npairs, n), dtype=np.uint64)
    for sign, host in ((+1, host_pos), (-1, host_neg)):
        assert L.lgc_p1_mask(ph._h, cols.ctypes.data_as(C.c_void_p), npairs, dV, sign, dOut) == 0
        assert L.lgc_dev_download(out.ctypes.data_as(C.c_void_p), dOut, nbytes) == 0
        assert not (out >> np.uint64(32)).any()                  # what the mapped peer reads: 32-bit words only
        assert np.array_equal(out, host)
    sub = np.zeros(npairs, dtype=np.uint64); shares = np.zeros(npairs, dtype=np.uint64)
    assert L.lgc_p1_ti_a_batch(ph._h, cols.ctypes.data_as(C.c_void_p), npairs, dV, dIn, sub.ctypes.data_as(C.c_void_p), dOut,
                               shares.ctypes.data_as(C.c_void_p)) == 0
    assert L.lgc_dev_download(out.ctypes.data_as(C.c_void_p), dOut, nbytes) == 0
    assert not (out >> np.uint64(32)).any()
    assert np.array_equal(out, host_neg)                         # a - y, the same message as mask(-1)
    exp = [(int((inn[q].astype(object) * V[q].astype(object)).sum()) & 0xffffffff) for q in range(npairs)]
    assert shares.tolist() == exp
    for ptr in (dV, dIn, dOut):
        L.lgc_dev_free(ptr)
    ph.close()


@pytest.mark.gpu
def test_devices_preflight_names_the_missing_index(lgc):
    """bin/linreg --devices / bench.py --gpus N check their device list before anything is allocated (lgc_devices_preflight)"""
    n = lgc.device_count()
    lgc.devices_preflight([0])
    lgc.devices_preflight([0, 0, 0])                    # an index may repeat: several blocks on one GPU
    with pytest.raises(lgc.LgcError) as e:
        lgc.devices_preflight([0, n])
    assert "device index %d does not exist" % n in str(e.value) and e.value.code == -1
    with pytest.raises(lgc.LgcError):
        lgc.devices_preflight([])
    if n >= 2:
        lgc.devices_preflight(list(range(n)))           # one xGMI hive: every pair reachable
