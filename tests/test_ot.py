"""IKNP OT extension: CPU mirror self-consistency (no GPU) and GPU vs mirror (bit-exact transcripts)."""
import numpy as np
import pytest


def _setup(rng):
    seeds0 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
    seeds1 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
    delta = rng.integers(0, 256, size=16, dtype=np.uint8)
    dbits = np.unpackbits(delta, bitorder="little")
    seeds_s = np.where(dbits[:, None] == 1, seeds1, seeds0)      # what the base OTs give the sender
    return seeds0, seeds1, delta, seeds_s


def _ip(a, b, w):
    m = (1 << w) - 1
    return [sum(int(x) * int(y) for x, y in zip(ra, rb)) & m for ra, rb in zip(a, b)]


@pytest.mark.parametrize("w", [64, 32])
def test_cpu_mirror_gilboa_and_labels(gccpu, w):
    rng = np.random.default_rng(w)
    seeds0, seeds1, delta, _ = _setup(rng)
    npairs, n = 3, 5
    mask = np.uint64((1 << w) - 1) if w < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    a = rng.integers(0, 2 ** 63, size=(npairs, n), dtype=np.uint64) & mask
    b = rng.integers(0, 2 ** 63, size=(npairs, n), dtype=np.uint64) & mask
    m = npairs * n * w
    cb = a.view(np.uint8) if w == 64 else a.astype(np.uint32).view(np.uint8)
    u, rt, rq = gccpu.iknp_extend(seeds0, seeds1, delta.tobytes(), cb, m, 0)
    y, ss, sr = gccpu.iknp_gilboa(rt, rq, delta.tobytes(), a, b, w, 0)
    got = [(int(x) + int(z)) & ((1 << w) - 1) for x, z in zip(ss, sr)]
    assert got == _ip(a, b, w)
    # labels
    mm = 300
    choice = rng.integers(0, 2, size=mm, dtype=np.uint8)
    m0 = rng.integers(0, 256, size=(mm, 16), dtype=np.uint8); m1 = rng.integers(0, 256, size=(mm, 16), dtype=np.uint8)
    u, rt, rq = gccpu.iknp_extend(seeds0, seeds1, delta.tobytes(), np.packbits(choice, bitorder="little"), mm, 7)
    e, out = gccpu.iknp_labels(rt, rq, delta.tobytes(), choice, m0, m1, 1000)
    assert np.array_equal(out, np.where(choice[:, None] == 1, m1, m0))


@pytest.mark.gpu
@pytest.mark.parametrize("w,npairs,n", [(64, 3, 17), (32, 2, 50), (64, 1, 1000)])
def test_gpu_gilboa_matches_mirror(lgc, gccpu, w, npairs, n):
    rng = np.random.default_rng(w + n)
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    mask = np.uint64((1 << w) - 1) if w < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    ctr = tw = 0
    for rep in range(2):                      # two transfers on one session: stream positions advance
        a = rng.integers(0, 2 ** 63, size=(npairs, n), dtype=np.uint64) & mask
        b = rng.integers(0, 2 ** 63, size=(npairs, n), dtype=np.uint64) & mask
        u = R.gilboa_start(a, w)
        y, ss = S.gilboa(b, w, u)
        sr = R.gilboa_finish(y)
        assert [(int(x) + int(z)) & ((1 << w) - 1) for x, z in zip(ss, sr)] == _ip(a, b, w)
        m = npairs * n * w
        cb = a.view(np.uint8) if w == 64 else a.astype(np.uint32).view(np.uint8)
        cu, rt, rq = gccpu.iknp_extend(seeds0, seeds1, delta.tobytes(), cb, m, ctr)
        cy, css, csr = gccpu.iknp_gilboa(rt, rq, delta.tobytes(), a, b, w, tw)
        assert np.array_equal(u, cu)
        assert np.array_equal(y, cy) and np.array_equal(ss, css) and np.array_equal(sr, csr)
        ctr += (m + 127) // 128
        tw += m
    S.close(); R.close()


@pytest.mark.gpu
def test_gpu_gilboa_receives_in_flight(lgc):
    """several receives started before the first is finished (the pipelined OT mode of the host):
    finishes complete the oldest batch; the sender answers in order"""
    rng = np.random.default_rng(9)
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    w = 64
    shapes = [(2, 40), (1, 300), (3, 7)]
    A = [rng.integers(0, 2 ** 63, size=sh, dtype=np.uint64) for sh in shapes]
    B = [rng.integers(0, 2 ** 63, size=sh, dtype=np.uint64) for sh in shapes]
    us = [R.gilboa_start(a, w) for a in A]                     # three in flight
    for a, b, u in zip(A, B, us):
        y, ss = S.gilboa(b, w, u)
        sr = R.gilboa_finish(y)
        assert [(int(x) + int(z)) & (2 ** 64 - 1) for x, z in zip(ss, sr)] == _ip(a, b, w)
    with pytest.raises(lgc.LgcError):                           # nothing in flight any more
        R.gilboa_finish(np.zeros(64, dtype=np.uint64))
    for a in A + [A[0]]:
        R.gilboa_start(a, w)                                    # four in flight is the limit
    with pytest.raises(lgc.LgcError):
        R.gilboa_start(A[0], w)
    S.close(); R.close()


@pytest.mark.gpu
def test_gpu_label_ot_matches_mirror(lgc, gccpu):
    rng = np.random.default_rng(5)
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    m = 5150 * 4 + 3                          # ragged: not a multiple of 128
    choice = rng.integers(0, 2, size=m, dtype=np.uint8)
    m0 = rng.integers(0, 256, size=(m, 16), dtype=np.uint8); m1 = rng.integers(0, 256, size=(m, 16), dtype=np.uint8)
    u = R.labels_start(choice)
    e = S.labels(m0, m1, u)
    out = R.labels_finish(e)
    assert np.array_equal(out, np.where(choice[:, None] == 1, m1, m0))
    cu, rt, rq = gccpu.iknp_extend(seeds0, seeds1, delta.tobytes(), np.packbits(choice, bitorder="little"), m, 0)
    ce, cout = gccpu.iknp_labels(rt, rq, delta.tobytes(), choice, m0, m1, 0)
    assert np.array_equal(u, cu) and np.array_equal(e, ce)
    S.close(); R.close()


@pytest.mark.gpu
def test_gpu_phase1_ot_mode_shares(lgc, oracle):
    """OT-mode phase 1 for two data providers (role rule of src/phase1.c:392): the GPU Gilboa
    shares plus the local blocks recombine to the oracle's aggregate"""
    rng = np.random.default_rng(9)
    n, d, p, w = 40, 4, 56, 64
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    yv = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
    Xq = oracle.quantize(X, p, n, w).reshape(n, d); yq = oracle.quantize(yv, p, n, w)
    A, b = oracle.aggregate(Xq, yq, n, d, p, w)
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    # DPs 0 (cols 0,1) and 1 (cols 2,3 + y): different parity -> the higher index is the OT sender
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    U = lambda v: np.ascontiguousarray(v).astype(np.int64).view(np.uint64)
    pairs = [(i, j) for i in (2, 3) for j in (0, 1)]       # sender rows i (party 1), receiver cols j (party 0)
    a = np.stack([U(Xq[:, j]) for (_, j) in pairs] + [U(Xq[:, j]) for j in (0, 1)])       # receiver values
    bv = np.stack([U(Xq[:, i]) for (i, _) in pairs] + [U(yq), U(yq)])                     # sender values (target rows last)
    u = R.gilboa_start(a, w)
    y, ss = S.gilboa(bv, w, u)
    sr = R.gilboa_finish(y)
    m = (1 << 64) - 1
    for q, (i, j) in enumerate(pairs):
        assert (int(ss[q]) + int(sr[q])) & m == int(A[oracle.lib.orc_idx(i, j)])
    for t, j in enumerate((0, 1)):
        assert (int(ss[4 + t]) + int(sr[4 + t])) & m == int(b[j])
    S.close(); R.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,p", [(64, 56), (32, 30)])
def test_gpu_gilboa_mid_size_against_semantic_oracle(lgc, oracle, w, p):
    """Gilboa inner products at n = 1000, 32 pairs (2.0e6 / 1.0e6 extended OTs in one batch, several trips of every
    kernel loop, grid.y striding) against the SEMANTIC oracle -- the wrap-around inner products of the quantised
    columns that src/phase1.c:43-96 shares -- not against the CPU mirror of the protocol: share_sender + share_receiver
    == <a_q, b_q> mod 2^w for every pair, at both widths."""
    rng = np.random.default_rng(1000 + w)
    n, npairs, d = 1000, 32, 8
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    Xq = oracle.quantize(X, p, n, w).reshape(n, d)
    mask = (1 << w) - 1
    U = lambda v: (np.ascontiguousarray(v).astype(np.int64).view(np.uint64) & np.uint64(mask))
    cols = [(int(rng.integers(0, d)), int(rng.integers(0, d))) for _ in range(npairs)]
    a = np.stack([U(Xq[:, i]) for i, _ in cols]); b = np.stack([U(Xq[:, j]) for _, j in cols])
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    for rep in range(2):                                   # a second batch on the same session: stream / tweak counters advance
        u = R.gilboa_start(a, w)
        y, ss = S.gilboa(b, w, u)
        sr = R.gilboa_finish(y)
        for q, (i, j) in enumerate(cols):
            exp = oracle.inner_product_wrap(Xq[:, i], Xq[:, j], w) if hasattr(oracle, "inner_product_wrap") else \
                (sum(int(x) * int(z) for x, z in zip(Xq[:, i].tolist(), Xq[:, j].tolist())) & mask)
            assert (int(ss[q]) + int(sr[q])) & mask == exp, (w, rep, q)
        assert len(set(int(v) for v in ss)) > npairs // 2     # the shares are masks, not the answers
    S.close(); R.close()


def _openssl_aes_ctr(key, first_block, nblocks):
    """AES-128-CTR keystream written over OpenSSL's block function: block c = AES_key(c as a little-endian 128-bit number)"""
    import ctypes
    import ctypes.util
    crypto = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
    sched = ctypes.create_string_buffer(256)                           # AES_KEY
    assert crypto.AES_set_encrypt_key(bytes(key), 128, sched) == 0
    out = bytearray()
    inb, outb = ctypes.create_string_buffer(16), ctypes.create_string_buffer(16)
    for c in range(first_block, first_block + nblocks):
        ctypes.memmove(inb, int(c).to_bytes(16, "little"), 16)
        crypto.AES_encrypt(inb, outb, sched)
        out += outb.raw
    return np.frombuffer(bytes(out), dtype=np.uint8)


@pytest.mark.gpu
def test_gpu_label_ot_against_openssl_and_numpy(lgc):
    """The IKNP extension of the GPU (ot.hip) against a restatement that shares NOTHING with the product or with
    oracle/gc_cpu.cpp (which is compiled from the product's headers): the column PRG is OpenSSL's AES-128 in counter mode
    under each base-OT seed, the 128 x m bit-matrix transpose is numpy, the correlation-robust hash is
    helpers.openssl_gate_hash (the gate hash from its definition over OpenSSL).  A 256-OT batch and a second, ragged one on
    the same session (the PRG counter and the tweak counter advance): the receiver's message u, the sender's ciphertext
    pairs e and the receiver's output labels, byte for byte.  Protocol as the reference's call sites use it
    (src/input.c:28-44: one extended 1-of-2 OT per input bit, payloads = the two wire labels; src/phase1.c:58-65)."""
    from helpers import openssl_gate_hash
    rng = np.random.default_rng(77)
    seeds0, seeds1, delta, seeds_s = _setup(rng)
    dbits = np.unpackbits(delta, bitorder="little")
    S = lgc.OtSender(delta.tobytes(), seeds_s); R = lgc.OtReceiver(seeds0, seeds1)
    ctr = tw = 0
    for m in (256, 131):
        choice = rng.integers(0, 2, size=m, dtype=np.uint8)
        m0 = rng.integers(0, 256, size=(m, 16), dtype=np.uint8); m1 = rng.integers(0, 256, size=(m, 16), dtype=np.uint8)
        u = R.labels_start(choice)
        e = S.labels(m0, m1, u)
        out = R.labels_finish(e)
        # ---- the restatement
        m128 = (m + 127) // 128
        cb = np.zeros(m128 * 16, dtype=np.uint8)
        packed = np.packbits(choice, bitorder="little")
        cb[:len(packed)] = packed
        T = np.stack([_openssl_aes_ctr(seeds0[j], ctr, m128) for j in range(128)])          # receiver's columns t_j = G(k0_j)
        G1 = np.stack([_openssl_aes_ctr(seeds1[j], ctr, m128) for j in range(128)])
        U = T ^ G1 ^ cb[None, :]                                                            # u_j = G(k0_j) ^ G(k1_j) ^ c
        # sender: it holds k_{delta_j}, so q_j = G(k_{delta_j}) ^ delta_j u_j  ( = t_j ^ delta_j c )
        Gs = np.stack([_openssl_aes_ctr(seeds_s[j], ctr, m128) for j in range(128)])
        Q = Gs ^ (U * dbits[:, None])
        # bit-matrix transpose: row i holds bit i of every column, column j at bit j (LSB first in both directions)
        tb = np.unpackbits(T, axis=1, bitorder="little")[:, :m]                             # (128, m)
        qb = np.unpackbits(Q, axis=1, bitorder="little")[:, :m]
        rows_t = np.packbits(tb.T, axis=1, bitorder="little")                               # (m, 16)
        rows_q = np.packbits(qb.T, axis=1, bitorder="little")
        assert np.array_equal(rows_q, rows_t ^ (delta[None, :] * choice[:, None]))          # the IKNP correlation itself
        tweaks = np.arange(tw, tw + m, dtype=np.uint64)
        e0 = m0 ^ openssl_gate_hash(rows_q, tweaks)
        e1 = m1 ^ openssl_gate_hash(rows_q ^ delta[None, :], tweaks)
        ht = openssl_gate_hash(rows_t, tweaks)
        exp_out = np.where(choice[:, None] == 1, e1, e0) ^ ht
        # ---- against the GPU
        assert np.array_equal(np.asarray(u, dtype=np.uint8).reshape(-1), U.reshape(-1)), m
        ge = np.asarray(e, dtype=np.uint8).reshape(m, 2, 16)
        assert np.array_equal(ge[:, 0], e0) and np.array_equal(ge[:, 1], e1), m
        assert np.array_equal(np.asarray(out, dtype=np.uint8).reshape(m, 16), exp_out), m
        assert np.array_equal(exp_out, np.where(choice[:, None] == 1, m1, m0))              # and the chosen labels are delivered
        ctr += m128
        tw += m
    S.close(); R.close()
