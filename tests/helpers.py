"""shared helpers for the parity tests"""
import numpy as np


def synth_system(oracle, rng, n, d, w, p, lam=0.001, sigma=0.1):
    """experiments/generate_tests.py:159-169 distribution, quantised and aggregated
    by the oracle.  Returns (A_total, b_total) as uint64 (T and d entries)."""
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    beta = rng.random(d)
    y = X @ beta + sigma * rng.standard_normal(n)
    Xq = oracle.quantize(X, p, n, w); yq = oracle.quantize(y, p, n, w)
    A, b = oracle.aggregate(Xq, yq, n, d, p, w)
    return A, b


def split_shares(rng, A, b, nshares, w):
    """additive shares mod 2^w of the (T + d) vector; share-major (nshares, T + d)"""
    tot = np.concatenate([A, b]).astype(np.uint64)
    m = np.uint64((1 << w) - 1) if w < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sh = rng.integers(0, 2 ** 63, size=(nshares, tot.size), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(nshares, tot.size), dtype=np.uint64)
    sh &= m
    with np.errstate(over="ignore"):
        sh[0] = (tot - sh[1:].sum(axis=0, dtype=np.uint64)) & m
    return sh


def oracle_solve(oracle, A, b, d, w, p, alg, iters, lam, normalize, trace=False):
    a = oracle.sum_shares(np.asarray(A, dtype=np.uint64)[None, :], w)
    bb = oracle.sum_shares(np.asarray(b, dtype=np.uint64)[None, :], w)
    if normalize:
        a, bb = oracle.circuit_input(a, bb, d, lam, p, w)
    if alg == "cgd":
        return oracle.cgd(a, bb, d, p, w, iters, trace=trace), a, bb
    if alg == "cholesky":
        return oracle.cholesky(a, bb, d, p, w), a, bb
    return oracle.ldlt(a, bb, d, p, w), a, bb


def sx(v, w):
    v = np.asarray(v, dtype=np.uint64)
    if w == 32:
        return v.astype(np.uint32).astype(np.int32).astype(np.int64)
    return v.astype(np.int64)


def openssl_gate_hash(labels, tweaks):
    """The fixed-key AES gate hash H(x, t) = AES_k(sigma(x) ^ t) ^ sigma(x) ^ t, sigma(xL || xR) = (xL ^ xR) || xL, written
    here from its definition over OpenSSL's AES-128 (libcrypto through ctypes): shares no line with the product's T-table AES
    or with oracle/gc_cpu.cpp (which is compiled from the product's headers), so agreement is an independent check of both.
    labels (n, 16) uint8, tweaks (n,) uint64 in the low half.  Key: FIPS-197 Appendix B (the build's fixed public key)."""
    import ctypes
    import ctypes.util
    import numpy as np
    crypto = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    sched = ctypes.create_string_buffer(256)                           # AES_KEY
    assert crypto.AES_set_encrypt_key(key, 128, sched) == 0
    x = np.ascontiguousarray(labels, dtype=np.uint8).reshape(-1, 16)
    out = np.zeros_like(x)
    inb, outb = ctypes.create_string_buffer(16), ctypes.create_string_buffer(16)
    for i in range(len(x)):
        # the label is a little-endian 128-bit number: xL = its HIGH half (bytes 8..15), xR = its low half (bytes 0..7);
        # sigma(x) = (xL ^ xR) || xL has xL ^ xR in the high half and xL in the low half; the 64-bit tweak goes into the low half
        xl, xr = x[i, 8:], x[i, :8]
        k = np.concatenate([xl, xl ^ xr])
        tw = np.frombuffer(int(tweaks[i]).to_bytes(8, "little"), dtype=np.uint8)
        k[:8] ^= tw
        ctypes.memmove(inb, k.tobytes(), 16)
        crypto.AES_encrypt(inb, outb, sched)
        out[i] = np.frombuffer(outb.raw, dtype=np.uint8) ^ k
    return out


def free_ports(k):
    """k listening ports BELOW the kernel's ephemeral range (ip_local_port_range, 32768-60999 here): a port from bind(0)
    lies inside that range, and while its party is not listening yet a peer's connect() attempt can be given the same number as
    its SOURCE port -- the attempt then connects to itself (TCP simultaneous open) or the party's bind fails, and the run hangs
    (one such hang in ~600 runs of the GPU suite)"""
    import socket, random
    lo = 12000
    try:
        hi = min(30000, int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0]) - 1)
    except (OSError, ValueError):
        hi = 30000
    ports, rnd = [], random.SystemRandom()
    while len(ports) < k:
        p = rnd.randrange(lo, hi)
        if p in ports:
            continue
        s_ = socket.socket()
        try:
            s_.bind(("127.0.0.1", p))
            ports.append(p)
        except OSError:
            pass
        finally:
            s_.close()
    return ports
