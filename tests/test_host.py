"""Host side (C): wire format and config parser on CPU; the five-process `bin/linreg` run of the
README example (BASELINE.json config 1) on the GPU box."""
import ctypes as C
import os
import re
import socket
import subprocess
import time
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "linreg-mpc_amd", "host")


@pytest.fixture(scope="module")
def hostlib():
    so = os.path.join(HOST, "libhosttest.so")
    subprocess.check_call(["make", "-C", HOST, "libhosttest.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.pmsg_packed_size.restype = C.c_size_t
    L.pmsg_packed_size.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    L.pmsg_pack.restype = C.c_size_t
    L.pmsg_pack.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
    L.pmsg_unpack.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    return L


def _pack(L, vec, value):
    v = np.ascontiguousarray(vec, dtype=np.uint64)
    n = L.pmsg_packed_size(v.ctypes.data, len(v), value)
    out = np.zeros(n, dtype=np.uint8)
    assert L.pmsg_pack(v.ctypes.data, len(v), value, out.ctypes.data) == n
    return bytes(out)


def test_pmsg_wire_format(hostlib):
    # proto2: repeated uint64 vector = 1 [packed]; required uint64 value = 2 (src/protobuf/*.proto)
    assert _pack(hostlib, [1, 300], 5) == bytes.fromhex("0a0301ac021005")
    assert _pack(hostlib, [], 0) == bytes.fromhex("1000")
    big = _pack(hostlib, [2 ** 64 - 1], 2 ** 63)
    assert big == bytes.fromhex("0a0a" + "ff" * 9 + "01" + "10" + "80" * 9 + "01")
    rng = np.random.default_rng(0)
    vec = rng.integers(0, 2 ** 63, size=1000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    blob = _pack(hostlib, vec, 12345678901234567890)
    pv = C.POINTER(C.c_uint64)(); n = C.c_size_t(); val = C.c_uint64()
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    assert hostlib.pmsg_unpack(buf.ctypes.data, len(buf), C.byref(pv), C.byref(n), C.byref(val)) == 0
    assert n.value == 1000 and val.value == 12345678901234567890
    assert np.array_equal(np.ctypeslib.as_array(pv, (1000,)), vec)
    # every varint length (1..10 bytes), every tail position of the word-at-a-time fast path
    def enc(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7f) | 0x80); v >>= 7
        out.append(v)
        return bytes(out)
    for n_el in list(range(0, 40)) + [63, 64, 65, 257, 1000]:
        bits = rng.integers(0, 65, size=n_el)
        vals = [int(rng.integers(0, 2 ** 63, dtype=np.uint64)) * 2 + 1 for _ in range(n_el)]
        vals = [v & ((1 << int(b)) - 1) for v, b in zip(vals, bits)]
        v2 = int(rng.integers(0, 2 ** 63, dtype=np.uint64))
        payload = b"".join(enc(v) for v in vals)
        expect = (b"\x0a" + enc(len(payload)) + payload if n_el else b"") + b"\x10" + enc(v2)
        blob = _pack(hostlib, np.array(vals, dtype=np.uint64), v2)
        assert blob == expect, n_el
        buf = np.frombuffer(blob, dtype=np.uint8).copy()
        assert hostlib.pmsg_unpack(buf.ctypes.data, len(buf), C.byref(pv), C.byref(n), C.byref(val)) == 0
        assert n.value == n_el and val.value == v2
        if n_el:
            assert np.ctypeslib.as_array(pv, (n_el,)).tolist() == vals
    # an over-long varint (11 continuation bytes) is rejected by both decoders
    for pad in (0, 100):
        bad = bytes.fromhex("0a") + enc(pad + 11) + b"\x01" * pad + b"\xff" * 10 + b"\x01" + bytes.fromhex("1000")
        bad = np.frombuffer(bad, dtype=np.uint8).copy()
        assert hostlib.pmsg_unpack(bad.ctypes.data, len(bad), C.byref(pv), C.byref(n), C.byref(val)) != 0
    # a message without the required `value` is rejected (recv_pmsg's check, phase1.c:112)
    bad = np.frombuffer(bytes.fromhex("0a0101"), dtype=np.uint8).copy()
    assert hostlib.pmsg_unpack(bad.ctypes.data, len(bad), C.byref(pv), C.byref(n), C.byref(val)) != 0


from helpers import free_ports as _free_ports    # ports below the ephemeral range (see there)


def _rewrite_ports(src, dst):
    tok = open(src).read().split("\n")
    n, d, P = map(int, tok[0].split())
    ports = _free_ports(P + 2)
    for i in range(P + 2):
        parts = tok[1 + i].split()
        parts[0] = "127.0.0.1:%d" % ports[i]
        tok[1 + i] = " ".join(parts)
    open(dst, "w").write("\n".join(tok))
    return P


def _run_all(infile, P, args, timeout=300, exe_name="linreg", env=None):
    exe = os.path.join(HOST, "bin", exe_name)
    procs = []
    for party in range(1, P + 3):
        cmd = [exe, infile, args[0], str(party)] + args[1:]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join("party %d rc %s: %s" % (k + 1, p.returncode, e.decode()[-600:])
                                                            for k, (p, (o, e)) in enumerate(zip(procs, outs)))
    return [o.decode() for o, _ in outs]


README_RESULT = ["0.984331027786964", "0.792399824970372", "0.754117840176144", "0.592849130685193", "0.057351715952213"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--use_ot"], ["--table_ring"], ["--ot_ring", "--table_ring"], ["--ti_ring", "--table_ring"],
                                   ["--table_lanes=4"], ["--input_ring"], ["--ti_ring", "--input_ring", "--table_ring"]],
                         ids=["ti", "ot", "table-ring", "ot-ring", "ti-ring", "table-lanes", "input-ring", "all-rings"])
def test_five_process_readme_example(tmp_path, golden_dir, extra):
    """bin/linreg examples/readme_example.in 56 $party cgd 10 0.001 (README.md:81) -> README.md:87"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    outs = _run_all(infile, P, ["56", "cgd", "10", "0.001"] + extra)
    ev = outs[1]
    last = ev.strip().splitlines()[-1]
    assert last.startswith("Result:")
    assert re.findall("-?[0-9]+\\.[0-9]+", last) == README_RESULT
    assert '{"n":"10", "d":"5" "p":"4"}' in ev and "Algorithm: cgd" in ev and "Number of gates:" in ev
    assert ev.count("Iteration") >= 30 and "Time elapsed:" in ev
    for k, o in enumerate(outs):
        assert "Party %d finished phase 1" % (k + 1) in o


@pytest.mark.gpu
@pytest.mark.parametrize("odd", [["--prec_phase2=50"]], ids=["precision"])
def test_an_option_given_to_one_party_only_is_an_error_not_a_wrong_result(tmp_path, golden_dir, odd):
    """the CSP and the Evaluator compare the fingerprints of their programs before the first table moves
    (host/tables.c: programs_agree): party 1 alone gets the extra option"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    exe = os.path.join(HOST, "bin", "linreg")
    procs = []
    for party in range(1, P + 3):
        cmd = [exe, infile, "56", str(party), "cholesky", "0", "0.001"] + (odd if party == 1 else [])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [q.communicate(timeout=120) for q in procs]
    assert procs[0].returncode != 0 and procs[1].returncode != 0
    assert b"built different programs" in outs[0][1] and b"built different programs" in outs[1][1]
    assert b"Result:" not in outs[1][0]


@pytest.mark.gpu
@pytest.mark.parametrize("alg,iters,async_ring", [("cgd", "10", "1"), ("cholesky", "0", "1"), ("cholesky", "0", "0"), ("cgd", "10", "2"), ("cholesky", "0", "2")])
def test_five_process_ring_with_the_asynchronous_garbler(tmp_path, golden_dir, oracle, alg, iters, async_ring):
    """--table_ring with the CSP's launches enqueued asynchronously (host/tables.c: a second thread sends the tokens; the
    library runs the table passes of critical-path launches on a stream of their own, two stashes in turn) -- the path
    programs of a thousand launches and more take by themselves, forced here on the README example (LINREG_RING_ASYNC=1),
    the synchronous loop forced the other way (0), and the asynchronous path on ONE stream (2): same Result line"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    outs = _run_all(infile, P, ["56", alg, iters, "0.001", "--table_ring", "--input_ring"], env=dict(os.environ, LINREG_RING_ASYNC=async_ring),
                    exe_name="linreg_testhooks")      # the production binary picks the form by the program's length alone
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    beta = oracle.linreg_file(os.path.join(golden_dir, "readme_example.in"), 56, -1, 64, 64, {"cholesky": 0, "cgd": 2}[alg], int(iters), 0.001)
    assert got == ["%.15f" % (int(b) / 2.0 ** 56) for b in beta]
    if alg == "cgd":
        assert got == README_RESULT


@pytest.mark.gpu
def test_five_process_cholesky_matches_oracle(tmp_path, golden_dir, oracle):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    outs = _run_all(infile, P, ["56", "cholesky", "0", "0.001"])
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    beta = oracle.linreg_file(os.path.join(golden_dir, "readme_example.in"), 56, -1, 64, 64, 0, 0, 0.001)
    assert got == ["%.15f" % (int(b) / 2.0 ** 56) for b in beta]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--use_ot"]], ids=["ti", "ot"])
def test_secure_multiplication_binary(tmp_path, golden_dir, extra):
    """phase-1-only benchmark driver: the JSON lines of src/cmd/secure_multiplication.c:74,98-104"""
    import json
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    exe = os.path.join(HOST, "bin", "secure_multiplication")
    procs = [subprocess.Popen([exe, infile, "56", str(k)] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for k in range(1, P + 3)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-1000:]
    assert '{"n":"10", "d":"5", "p":"3"}' in outs[0][0].decode()
    for k, (o, _) in enumerate(outs):
        lines = [l for l in o.decode().splitlines() if l.startswith('{"party"')]
        t = json.loads(lines[0])
        assert t["party"] == str(k + 1) and float(t["realtime"]) >= 0 and (k == 1 or float(t["realtime"]) > 0 or extra)
        sent = json.loads(lines[1])["bytes_sent"]
        sends = json.loads(lines[1])["sends"]
        assert len(sent) == P + 2 and len(sends) == P + 2
        assert all((b > 0) == (c > 0) for b, c in zip(sent, sends))
        if k >= 2:
            assert sum(sent) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("alg,w,p,ring", [("cgd", 64, 56, 0), ("cholesky", 64, 54, 0), ("ldlt", 32, 28, 0),
                                          ("cgd", 64, 56, 2), ("cholesky", 32, 28, 1), ("cgd", 64, 56, -3), ("ldlt", 64, 56, -1)])
def test_test_linear_system_binary(tmp_path, oracle, alg, w, p, ring):
    """two-party phase-2 benchmark (src/cmd/test/test_linear_system.c) vs the oracle on the same file;
    ring > 0: the garbled tables stay in a device-resident ring shared between the two processes;
    ring < 0: they cross the network striped over -ring extra connections (--table_lanes)"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(17)
    d, iters = 4, 5
    X = rng.standard_normal((40, d)); X /= np.abs(X).max(axis=0)
    A = X.T @ X / (40 * d) + np.eye(d) * 1e-2
    sol = rng.random(d)
    b = A @ sol
    path = str(tmp_path / "ls.in")
    with open(path, "w") as f:                       # experiments/generate_tests.py:9-26 layout
        f.write("%d %d\n" % (d, d))
        for i in range(d):
            f.write(" ".join(repr(float(v)) for v in A[i]) + " \n")
        f.write("%d\n" % d + " ".join(repr(float(v)) for v in b) + " \n")
        f.write("%d\n" % d + " ".join(repr(float(v)) for v in sol) + " ")
    port = _free_ports(1)[0]
    exe = os.path.join(HOST, "bin", "test_linear_system")
    opt = ["--width=%d" % w, "--host=127.0.0.1"] + (["--table_ring=%d" % ring] if ring > 0 else []) + (["--table_lanes=%d" % -ring] if ring < 0 else [])
    procs = [subprocess.Popen([exe, str(port), str(k), path, alg, str(iters), str(p)] + opt,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in (1, 2)]
    outs = [q.communicate(timeout=300) for q in procs]
    for q, (o, e) in zip(procs, outs):
        assert q.returncode == 0, e.decode()[-1000:]
    ev = outs[1][0].decode()
    got = re.findall("-?[0-9]+\\.[0-9]+", ev.strip().splitlines()[-1])
    aq = np.array([oracle.lib.orc_double_to_fixed(float(A[i, j]), p, w) for i in range(d) for j in range(i + 1)], dtype=np.int64)
    bq = np.array([oracle.lib.orc_double_to_fixed(float(v), p, w) for v in b], dtype=np.int64)
    if alg == "cgd":
        exp = oracle.cgd(aq, bq, d, p, w, iters)
    elif alg == "cholesky":
        exp = oracle.cholesky(aq, bq, d, p, w)
    else:
        exp = oracle.ldlt(aq, bq, d, p, w)
    assert got == ["%.15f" % (int(v) / 2.0 ** p) for v in exp]
    assert "Number of gates:" in ev and "Time elapsed:" in ev and "Algorithm: %s" % alg in ev
    if alg != "ldlt":                                 # sanity: close to the floating-point solution
        assert np.allclose([float(x) for x in got], sol, atol=1e-3 if alg == "cgd" else 1e-6)
    # the experiment driver's view of the same stdout (experiments/test_phase2_aws.py:70-186)
    import results
    run = results.parse_exec(ev, alg)
    assert run["gate_count"] > 0 and run["time"] > 0 and len(run["result"]) == d
    if alg == "cgd":
        g, t = run["iter_gates"], run["iter_times"]
        assert len(g) == len(t) == len(run["iter_solutions"]) == iters
        assert all(b > a for a, b in zip(g, g[1:])) and g[-1] <= run["gate_count"]
        assert len(set(np.diff(g))) == 1              # every iteration is the same circuit
        assert all(b >= a for a, b in zip(t, t[1:])) and t[-1] <= run["time"]
        assert run["iter_solutions"][-1] == run["result"]
    out = str(tmp_path / "ls.out")
    err = results.write_phase2_out(out, 40, d, alg, run, sol, condition_number=float(np.linalg.cond(A)))
    rows = open(out).read().split("\n")
    # this build's gate count, and next to it the reference's count for the same solve (none for ldlt)
    import linreg_gc
    refg = linreg_gc.reference_gate_count(alg, w, d, iters if alg == "cgd" else 0)
    assert run["ref_gate_count"] == refg and (refg is None) == (alg == "ldlt")
    assert rows[0] == "n d algorithm ot_time time error gate_count" + ("" if refg is None else " ref_gate_count")
    assert rows[1].split()[:3] == ["40", str(d), alg] and abs(float(rows[1].split()[5]) - err) < 1e-12
    if refg is not None:
        assert int(rows[1].split()[7]) == refg               # (not always the larger one: Kogge-Stone adders trade gates for depth)
    if alg == "cgd":
        assert rows[2] == "iter_i error_i obj_i time_i gate_count_i" and len(rows) == 3 + iters + 10
        assert int(rows[2 + iters].split()[4]) == run["gate_count"]


def test_config_parser_and_owner_map(hostlib, golden_dir):
    """header of the input file (reference src/config.c:24-44) and get_owner (src/phase1.c:25-33)"""
    class Cfg(C.Structure):
        _fields_ = [("party", C.c_int), ("num_parties", C.c_int), ("endpoint", C.POINTER(C.c_char_p)),
                    ("index_owned", C.POINTER(C.c_ssize_t)), ("n", C.c_size_t), ("d", C.c_size_t), ("input", C.c_void_p)]
    hostlib.config_new.argtypes = [C.POINTER(C.POINTER(Cfg)), C.c_char_p]
    hostlib.config_destroy.argtypes = [C.POINTER(C.POINTER(Cfg))]
    hostlib.config_owner.argtypes = [C.POINTER(Cfg), C.c_size_t]
    pc = C.POINTER(Cfg)()
    assert hostlib.config_new(C.byref(pc), os.path.join(golden_dir, "readme_example.in").encode()) == 0
    c = pc.contents
    assert (c.n, c.d, c.num_parties) == (10, 5, 5)
    assert [c.endpoint[i].decode() for i in range(5)] == ["localhost:%d" % p for p in range(1234, 1239)]
    assert [c.index_owned[i] for i in range(5)] == [-1, -1, 0, 1, 2]
    # columns 0 | 1 | 2,3,4 + target (row 5) -> 0-based party indices 2, 3, 4
    assert [hostlib.config_owner(pc, r) for r in range(6)] == [2, 3, 4, 4, 4, 4]
    hostlib.config_destroy(C.byref(pc))
    assert not pc
    bad = C.POINTER(Cfg)()
    assert hostlib.config_new(C.byref(bad), b"/nonexistent/file") != 0


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [[], ["--ti_ring"]], ids=["sockets", "ti-ring"])
def test_five_process_64_32_split_share_level(tmp_path, golden_dir, oracle, gccpu, ring):
    """BASELINE config 4's build: phase 1 in 64 bits, phase 2 in 32 bits (--prec_phase2).  Every
    share is shifted on its own (src/phase1.c:609-638), so the result depends on the TI's
    randomness: the TI seed is pinned and the oracle replays the same AES-CTR stream."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    src = os.path.join(golden_dir, "readme_example.in")
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(src, infile)
    p1, p2, lam, iters = 56, 24, 0.001, 6
    seed = bytes(range(0x40, 0x50))
    os.environ["LINREG_TI_SEED"] = seed.hex()
    try:
        outs = _run_all(infile, P, ["%d" % p1, "cgd", str(iters), str(lam), "--width_phase2=32", "--prec_phase2=%d" % p2] + ring,
                        exe_name="linreg_testhooks")      # the production binary ignores LINREG_TI_SEED
    finally:
        del os.environ["LINREG_TI_SEED"]
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    inp = oracle.read_input(src)
    n, d = inp["n"], inp["d"]
    Xq = oracle.quantize(inp["X"], p1, n, 32); yq = oracle.quantize(inp["y"], p1, n, 32)   # cast through the 32-bit fixed_t
    words = gccpu.ti_stream_words(seed, 0, 64 * (2 * n + 1), 64)
    sA, sb, used = oracle.ti_shares(Xq.reshape(n, d), yq, n, d, p1, 64, inp["start"], words)
    cA = np.stack([oracle.convert_shares(r, p1, p2, 64, 32) for r in sA])
    cb = np.stack([oracle.convert_shares(r, p1, p2, 64, 32) for r in sb])
    a, bb = oracle.circuit_input(oracle.sum_shares(cA, 32), oracle.sum_shares(cb, 32), d, lam, p2, 32)
    beta = oracle.cgd(a, bb, d, p2, 32, iters)
    assert got == ["%.15f" % (int(v) / 2.0 ** p2) for v in beta]
    # and the per-share shift really matters here: shifting the total instead gives another input
    tot = oracle.aggregate(Xq.reshape(n, d), yq, n, d, p1, 64)
    assert not np.array_equal(oracle.sum_shares(cA, 32), oracle.sum_shares(oracle.convert_shares(tot[0], p1, p2, 64, 32)[None, :], 32))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--table_ring"]], ids=["socket", "ring"])
def test_five_process_lambda_sweep(tmp_path, golden_dir, oracle, extra):
    """bin/linreg ... --lambdas=l1,l2,l3: CSP and Evaluator as separate processes run ONE merged program for
    all lambdas; the three data providers share their inputs (label OT with the CSP) once.  Every Result
    line equals the oracle's single-lambda run (src/linear.oc:52-57: lambda enters after the share sums)."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    src = os.path.join(golden_dir, "readme_example.in")
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(src, infile)
    lams = [0.001, 0.0, 0.25, 1e-6]
    outs = _run_all(infile, P, ["56", "cgd", "10", "123"] + ["--lambdas=" + ",".join(repr(l) for l in lams)] + extra)
    ev = outs[1].strip().splitlines()
    got = [(float(ev[i].split()[1]), re.findall("-?[0-9]+\\.[0-9]+", ev[i + 1])) for i in range(len(ev) - 1)
           if ev[i].startswith("Lambda:") and ev[i + 1].startswith("Result:")]
    assert [g[0] for g in got] == lams
    assert got[0][1] == README_RESULT                                 # lambda = 0.001 is the README run
    for lam, res in got:
        beta = oracle.linreg_file(src, 56, -1, 64, 64, 2, 10, lam)
        assert res == ["%.15f" % (int(b) / 2.0 ** 56) for b in beta], lam
    # the data providers ran exactly as in a single-lambda run
    for k in range(2, P + 2):
        assert "connected successfully to CSP and Evaluator" in outs[k]


@pytest.mark.gpu
@pytest.mark.parametrize("devs", ["0,0", "0,0,0", "0,1"], ids=["two-blocks-one-gpu", "three-blocks-one-gpu", "two-gpus"])
def test_five_process_lambda_sweep_on_several_devices(tmp_path, golden_dir, oracle, devs):
    """bin/linreg ... --lambdas=... --devices=g0,g1,... --table_ring: CSP and Evaluator each hold one block of the sweep
    per entry (one party object, one hipIpc ring, one thread each); block 0 garbles / evaluates the shared prefix and the
    other blocks take its share sums from device memory.  Host code is C, no torch.  Every Result line equals the
    oracle's single-lambda run, the stdout contract (Lambda: / Result: per value, in order) is the single-device one and
    the gate count is the one-program count (the prefix counted once).  "0,1" needs two GPUs (src/cmd/linreg.c:145-199
    runs one execYaoProtocol per circuit; src/linear.oc:31,52-57: lambda enters after the share sums)."""
    import linreg_gc
    if "1" in devs and linreg_gc.device_count() < 2:
        pytest.skip("needs two visible GPUs (the driver's multi-GPU node; one-GPU boxes skip)")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    src = os.path.join(golden_dir, "readme_example.in")
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(src, infile)
    lams = [0.001, 0.0, 0.25, 1e-6, 0.5]
    lam_opt = "--lambdas=" + ",".join(repr(l) for l in lams)
    outs = _run_all(infile, P, ["56", "cgd", "10", "123", lam_opt, "--table_ring", "--devices=" + devs])
    ev = outs[1].strip().splitlines()
    got = [(float(ev[i].split()[1]), re.findall("-?[0-9]+\\.[0-9]+", ev[i + 1])) for i in range(len(ev) - 1)
           if ev[i].startswith("Lambda:") and ev[i + 1].startswith("Result:")]
    assert [g[0] for g in got] == lams
    assert got[0][1] == README_RESULT
    for lam, res in got:
        beta = oracle.linreg_file(src, 56, -1, 64, 64, 2, 10, lam)
        assert res == ["%.15f" % (int(b) / 2.0 ** 56) for b in beta], lam
    gates = int(re.search("Number of gates: ([0-9]+)", outs[1]).group(1))
    one = _run_all(infile, P, ["56", "cgd", "10", "123", lam_opt, "--table_ring"])
    assert gates == int(re.search("Number of gates: ([0-9]+)", one[1]).group(1))
    for k in range(2, P + 2):                                          # the data providers ran exactly as in a single-lambda run
        assert "connected successfully to CSP and Evaluator" in outs[k]


@pytest.mark.gpu
def test_network_accounting_matches_profile_network(tmp_path):
    """-DPROFILE_NETWORK parity (experiments/test_phase1_aws.py:61, 245-252).  The published run
    experiments/results/phase1_network/test_LR_1000000x100_2_0_p{1..4}.out reads, per peer in party order,
        p1 (TI):  bytes [8, 24215018695, 24215000228]   flushes [3, 2551, 2551]
        p2:       bytes [12, 8, 0]                      flushes [4, 3, 1]
        p3 (DP1): bytes [4, 12, 24214997062]            flushes [2, 4, 2553]
        p4 (DP2): bytes [4, 4, 24215024180]             flushes [2, 2, 2554]
    with 2550 cross-party pairs: bytes per element 24215018695 / (2550 * 1e6) = 9.4961 (64-bit varints) and
    flushes = pairs + 1 (+ barrier flushes between neighbours).  Same structure here at n = 20000."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    n, d = 20000, 6
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    y = X @ rng.random(d)
    ports = _free_ports(4)
    path = str(tmp_path / "net.in")
    with open(path, "w") as f:
        f.write("%d %d 2\n127.0.0.1:%d\n127.0.0.1:%d\n127.0.0.1:%d 0\n127.0.0.1:%d 3\n%d %d\n" % (n, d, ports[0], ports[1], ports[2], ports[3], n, d))
        np.savetxt(f, X, fmt="%.17g")
        f.write("%d\n" % n)
        np.savetxt(f, y[None, :], fmt="%.17g")
    exe = os.path.join(HOST, "bin", "secure_multiplication")
    procs = [subprocess.Popen([exe, path, "56", str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in range(1, 5)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-1000:]
    pairs = 3 * 3 + 3                                   # columns 0-2 vs 3-5, and the target row vs columns 0-2
    byt, flu = [], []
    for o, _ in outs:
        t = o.decode()
        byt.append([int(v) for v in re.findall(r"Total bytes sent: (\d+)", t)])
        flu.append([int(v) for v in re.findall(r"Total flush done: (\d+)", t)])
    # control traffic: byte for byte the reference's
    assert byt[0][0] == 8 and byt[1] == [12, 8, 0] and byt[2][:2] == [4, 12] and byt[3][:2] == [4, 4]
    assert flu[0] == [3, pairs + 1, pairs + 1] and flu[1] == [4, 3, 1]
    assert flu[2] == [2, 4, pairs + 3] and flu[3] == [2, 2, pairs + 4]
    # payload: 64-bit values as varints, 9.4961 bytes per element in the reference's run
    ref = 24215018695 / (2550 * 1e6)
    for b in (byt[0][1], byt[0][2], byt[2][2], byt[3][2]):
        assert abs(b / (pairs * n) / ref - 1) < 1e-3, (b / (pairs * n), ref)


def _small_instance(path, n, d, starts, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    y = X @ rng.random(d) + 0.05 * rng.standard_normal(n)
    ports = _free_ports(len(starts) + 2)
    with open(path, "w") as f:
        f.write("%d %d %d\n127.0.0.1:%d\n127.0.0.1:%d\n" % (n, d, len(starts), ports[0], ports[1]))
        for k, st in enumerate(starts):
            f.write("127.0.0.1:%d %d\n" % (ports[2 + k], st))
        f.write("%d %d\n" % (n, d))
        np.savetxt(f, X, fmt="%.17g")
        f.write("%d\n" % n)
        np.savetxt(f, y[None, :], fmt="%.17g")


@pytest.mark.gpu
@pytest.mark.parametrize("starts,extra", [([0], []), ([0], ["--ti_ring"]), ([0], ["--use_ot"]),            # one provider: no cross-party pair at all
                                          ([0, 1, 2, 6], ["--ti_ring", "--table_ring"]),                  # ragged: 1, 1, 4, 1(+y) columns
                                          ([0, 1, 2, 6], ["--ot_ring"]), ([0, 1, 2, 6], [])],
                         ids=["P1-ti", "P1-ti-ring", "P1-ot", "ragged-ti-ring", "ragged-ot-ring", "ragged-ti"])
def test_edge_topologies(tmp_path, oracle, starts, extra):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    path = str(tmp_path / "edge.in")
    _small_instance(path, 37, 7, starts)
    outs = _run_all(path, len(starts), ["56", "cgd", "6", "0.01"] + extra)
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    beta = oracle.linreg_file(path, 56, -1, 64, 64, 2, 6, 0.01)
    assert got == ["%.15f" % (int(v) / 2.0 ** 56) for v in beta]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--ti_ring"], ["--ot_ring"]], ids=["ti", "ti-ring", "ot-ring"])
def test_phase1_in_32_bits(tmp_path, oracle, extra):
    """--width_phase1=32 --width_phase2=32 (the reference's BIT_WIDTH_32_P1 / _P2 builds), three providers"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    path = str(tmp_path / "w32.in")
    _small_instance(path, 25, 5, [0, 2, 3], seed=9)
    outs = _run_all(path, 3, ["24", "cholesky", "0", "0.01", "--width_phase1=32", "--width_phase2=32"] + extra)
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    beta = oracle.linreg_file(path, 24, -1, 32, 32, 0, 0, 0.01)
    assert got == ["%.15f" % (int(v) / 2.0 ** 24) for v in beta]


def test_host_parsers_under_address_sanitizer(tmp_path):
    """pmsg.c (wire message from a peer) and config.c (input-file header) compiled with ASan + UBSan and fed round trips,
    every truncation and mutated bytes, each input in a heap block of exactly its length (tests/tools/asan_host.c)"""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "asan_host")
    cc = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", HOST,
                         os.path.join(ROOT, "tests", "tools", "asan_host.c"), os.path.join(HOST, "pmsg.c"), os.path.join(HOST, "config.c"),
                         "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "cannot find" in cc.stderr and "san" in cc.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.rstrip().endswith("all ok"), (run.stdout[-1500:], run.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("victim,mark,extra,must_fail", [
    (2, "first table evaluated", ["--table_ring"], [1]),                 # the Evaluator dies while the tables stream through the ring
    (2, "first table evaluated", [], [1]),                               # ... through the socket pipeline
    (4, "input OT: u sent", ["--table_ring"], [1, 2]),                   # a data provider dies inside the label OT
    (1, "first table garbled", ["--table_ring"], [2]),                   # the CSP dies after its first table
], ids=["evaluator-ring", "evaluator-socket", "provider-label-ot", "csp-ring"])
def test_a_lost_party_makes_the_others_exit_nonzero_in_bounded_time(tmp_path, victim, mark, extra, must_fail):
    """the reference's convention is check() -> exit 1 for every party (src/check_error.h:5-9, src/cmd/linreg.c:206-211): when a
    party is lost mid-run, whoever still depends on it must fail within seconds -- not hang on a socket, a ring token or a
    device-side wait -- and nothing may be left behind.  bin/linreg_testhooks kills the victim at a named point of the
    protocol (LINREG_DIE_AT, a trace mark).  Parties that were already through with the victim may finish normally (the data
    providers leave after forwarding their labels, as in the reference)."""
    import signal, time
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    path = str(tmp_path / "lost.in")
    _small_instance(path, 60, 12, [0, 4, 8], seed=3)                      # three providers: parties 3, 4, 5
    exe = os.path.join(HOST, "bin", "linreg_testhooks")
    shm_before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    procs = []
    for party in range(1, 6):
        env = dict(os.environ, LINREG_TRACE="1")
        if party == victim:
            env["LINREG_DIE_AT"] = mark
        procs.append(subprocess.Popen([exe, path, "56", str(party), "cgd", "8", "0.01"] + extra, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, env=env, start_new_session=True))
    t0 = time.time()
    deadline = t0 + 60                                                    # start-up included (HIP runtime of five processes)
    died_at = None
    while time.time() < deadline and any(p.poll() is None for p in procs):
        if died_at is None and procs[victim - 1].poll() is not None:
            died_at = time.time()
        if died_at is not None and time.time() - died_at > 10:
            break
        time.sleep(0.02)
    hung = [k + 1 for k, p in enumerate(procs) if p.poll() is None]
    for p in procs:                                                       # whatever is still there: its whole process group
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
    outs = [p.communicate() for p in procs]
    assert procs[victim - 1].returncode == -signal.SIGKILL, (procs[victim - 1].returncode, outs[victim - 1][1].decode()[-400:])
    assert not hung, "parties %s still ran 10 s after party %d was lost; their last marks: %s" % (
        hung, victim, {k: outs[k - 1][1].decode()[-700:] for k in hung})
    for k in must_fail:
        assert procs[k - 1].returncode not in (0, None) and procs[k - 1].returncode > 0, \
            "party %d: rc %s, stderr %s" % (k, procs[k - 1].returncode, outs[k - 1][1].decode()[-400:])
    assert b"Result:" not in outs[1][0]                                   # no result without all parties
    # nothing left behind: every party reaped, no process of these groups alive, no new shared-memory files
    for p in procs:
        with pytest.raises(ProcessLookupError):
            os.killpg(p.pid, 0)
    if os.path.isdir("/dev/shm"):
        assert set(os.listdir("/dev/shm")) - shm_before == set()


@pytest.mark.gpu
def test_two_parties_with_different_dimensions_are_told_so_by_the_circuit(tmp_path):
    """src/linear.oc:109-114: the two parties' dimensions are compared INSIDE a circuit and the bit is revealed to both;
    when they differ both exit non-zero with the reference's message (bin/test_linear_system; rounds 1-3 compared on the
    socket).  With equal dimensions the gate count printed includes the comparison's 31 gates."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(3)
    paths = []
    for d in (3, 4):
        X = rng.standard_normal((30, d)); X /= np.abs(X).max(axis=0)
        A = X.T @ X / (30 * d) + np.eye(d) * 1e-2
        b = A @ rng.random(d)
        path = str(tmp_path / ("ls%d.in" % d))
        with open(path, "w") as f:
            f.write("%d %d\n" % (d, d))
            np.savetxt(f, A, fmt="%.17g")
            f.write("%d\n" % d)
            np.savetxt(f, b[None, :], fmt="%.17g")
            f.write("%d\n" % d)
            np.savetxt(f, np.zeros((1, d)), fmt="%g")
        paths.append(path)
    exe = os.path.join(HOST, "bin", "test_linear_system")
    port = _free_ports(1)[0]
    procs = [subprocess.Popen([exe, str(port), str(k), paths[k - 1], "cgd", "2", "56", "--host=127.0.0.1"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in (1, 2)]
    outs = [q.communicate(timeout=120) for q in procs]
    for q, (o, e) in zip(procs, outs):
        assert q.returncode == 1 and b"Inputs of the two parties differ." in e, (q.returncode, e.decode()[-300:])
    assert b"Result:" not in outs[1][0]
    port = _free_ports(1)[0]
    procs = [subprocess.Popen([exe, str(port), str(k), paths[0], "cgd", "2", "56", "--host=127.0.0.1"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in (1, 2)]
    outs = [q.communicate(timeout=120) for q in procs]
    assert all(q.returncode == 0 for q in procs), outs[0][1].decode()[-300:] + outs[1][1].decode()[-300:]
    import linreg_gc
    prog = linreg_gc.Program(linreg_gc.make_system(3, 64, 56, "cgd", 2, 0.0, 2, 0, 0, 1))
    assert int(re.search("Number of gates: ([0-9]+)", outs[1][0].decode()).group(1)) == prog.info.total_gates + 31


@pytest.mark.parametrize("party", [1, 3, 5], ids=["listener-only", "connects-and-listens", "connects-only"])
def test_a_party_whose_peers_never_appear_gives_up(tmp_path, golden_dir, party):
    """util_loop_connect (src/util.c:26-38) retries for ever and accept() waits for ever; with check() -> exit(1) as the
    failure convention (src/check_error.h:5-9) a party whose peer died before it was reachable would never end.  The
    connection phase has ONE deadline (LINREG_CONNECT_TIMEOUT seconds, default 300, 0 = for ever): exit 1 with the reason."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    infile = str(tmp_path / "readme.in")
    P = _rewrite_ports(os.path.join(golden_dir, "readme_example.in"), infile)
    assert party <= P + 2
    t0 = time.time()
    r = subprocess.run([os.path.join(HOST, "bin", "linreg"), infile, "56", str(party), "cgd", "10", "0.001"], capture_output=True,
                       text=True, timeout=60, env=dict(os.environ, LINREG_CONNECT_TIMEOUT="1.5"))
    assert r.returncode == 1 and time.time() - t0 < 20
    assert "LINREG_CONNECT_TIMEOUT" in r.stderr and "Could not create node" in r.stderr


def test_ports_for_the_runs_lie_below_the_ephemeral_range():
    lo = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    ports = _free_ports(8)
    assert len(set(ports)) == 8 and all(1024 < p < lo for p in ports)


def _parse_with(hostlib, path, n, d, c0, c1, own_y, prec, norm, w2, threads):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    hostlib.read_own_columns_threads.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_double,
                                                 C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    Xq = np.full(n * d, 7, dtype=np.int64); yq = np.full(n, 7, dtype=np.int64)      # the parser clears what it does not own
    f = libc.fopen(path.encode(), b"r")
    rc = hostlib.read_own_columns_threads(f, n, d, c0, c1, own_y, prec, norm, w2, Xq.ctypes.data, yq.ctypes.data, threads)
    libc.fclose(f)
    return rc, Xq.reshape(n, d), yq


def test_input_file_numbers_parsed_by_several_threads_are_the_ones_one_thread_reads(hostlib, tmp_path):
    """read_matrix / read_vector for a provider (src/linear.c:27-102): own columns converted as `%lf` would and quantised as
    (fixed_t)(v / normalizer * 2^p) (src/fixed.c:3-5), the rest zero; the scan is split over threads at token boundaries
    (host/readdata.c) -- every thread count gives the integers of a plain Python restatement, ragged white space, exponents,
    a mapped and a read buffer, both widths; malformed files are refused by every thread count"""
    rng = np.random.default_rng(11)
    for case, (n, d, c0, c1, own_y, prec, norm, w2) in enumerate([(37, 11, 3, 7, 1, 56, 1.0, 64), (64, 5, 0, 5, 0, 30, 8.0, 32),
                                                                    (300, 40, 39, 40, 1, 40, 3.0, 64), (2500, 210, 100, 200, 1, 56, 50000.0, 64)]):
        X = rng.standard_normal((n, d)) * 10.0 ** rng.integers(-3, 3, size=(n, d))
        y = rng.standard_normal(n)
        X[0, c0] = 1e300 if w2 == 64 else 3e9            # out of range: the integer-indefinite rule
        X[n - 1, c1 - 1] = -0.0
        seps = [" ", "  ", "\t", "\n", " \n", "\r\n"]
        path = str(tmp_path / ("m%d.in" % case))
        with open(path, "w") as f:
            f.write(" %d\t%d\n" % (n, d))
            for i in range(n):
                for j in range(d):
                    v = float(X[i, j])
                    tok = repr(v) if (i + j) % 3 else "%.17e" % v
                    f.write(tok + (seps[(i * d + j) % len(seps)] if case != 3 else (" " if j + 1 < d else "\n")))
            f.write("%d\n" % n + " ".join(repr(float(v)) for v in y))          # no white space after the last number
        def q(v):
            t = float(v) / norm * float(1 << prec)
            if w2 == 32:
                return int(t) if -2147483649.0 < t < 2147483648.0 else -2 ** 31
            return int(t) if -9223372036854775808.0 <= t < 9223372036854775808.0 else -2 ** 63
        expX = np.zeros((n, d), dtype=np.int64)
        for i in range(n):
            for j in range(c0, c1):
                expX[i, j] = q(X[i, j])
        expy = np.array([q(v) if own_y else 0 for v in y], dtype=np.int64)
        for threads in (1, 2, 3, 8, 64):
            rc, Xq, yq = _parse_with(hostlib, path, n, d, c0, c1, own_y, prec, norm, w2, threads)
            assert rc == 0, (case, threads)
            assert (Xq == expX).all() and (yq == expy).all(), (case, threads)
    # malformed: a count that does not match, a number missing, a token that is not a number in an own column, in a foreign one
    good = open(str(tmp_path / "m0.in")).read()
    n, d = 37, 11
    toks = good.split()
    def write(tl, name):
        pth = str(tmp_path / name)
        open(pth, "w").write(" ".join(tl) + "\n")
        return pth
    bad_files = [write(toks[:2] + toks[2:2 + n * d] + [str(n + 1)] + toks[3 + n * d:], "b0.in"),          # wrong length of y
                 write(toks[:-1], "b1.in"),                                                                # one number short
                 write(toks[:2 + 5] + ["1.5abc"] + toks[2 + 6:], "b2.in"),                                 # column 5: owned
                 write(toks[:2 + 1] + ["?"] + toks[2 + 2:], "b3.in"),                                      # column 1: foreign
                 write(["38", "11"] + toks[2:], "b4.in")]                                                  # header does not match
    for pth in bad_files:
        for threads in (1, 3, 8):
            rc, _, _ = _parse_with(hostlib, pth, n, d, 3, 7, 1, 56, 1.0, 64, threads)
            assert rc != 0, (pth, threads)


def test_sweep_plan_of_devices_0_1_is_the_plan_of_devices_0_0(lgc, hostlib):
    """host/sweep_plan.c: which circuits go to which --devices entry depends on the NUMBER of entries only -- the blocks (and
    with them every block's lowered program: records, launch list, gate-step numbers) of --devices=0,1 are those of
    --devices=0,0, which is what a one-GPU box can run.  Blocks are contiguous, cover the sweep, and never empty."""
    H = hostlib
    class Blk(C.Structure):
        _fields_ = [("device", C.c_int), ("lo", C.c_size_t), ("hi", C.c_size_t)]
    H.sweep_parse_devices.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int]; H.sweep_parse_devices.restype = C.c_int
    H.sweep_plan.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.c_int, C.POINTER(Blk)]; H.sweep_plan.restype = C.c_int

    def plan(text, n_lambdas):
        dev = (C.c_int * 16)()
        n = H.sweep_parse_devices(text.encode(), dev, 16)
        assert n > 0, text
        out = (Blk * 16)()
        k = H.sweep_plan(n_lambdas, dev, n, out)
        return [(out[i].device, out[i].lo, out[i].hi) for i in range(k)]

    for bad in ("", "0,", "a", "0,-1", "1,,2", ",".join(["0"] * 17)):
        assert H.sweep_parse_devices(bad.encode(), (C.c_int * 16)(), 16) == -1, bad
    for nl in (1, 2, 7, 8, 64):
        for text_a, text_b in (("0,1", "0,0"), ("0,1,2,3,4,5,6,7", "0,0,0,0,0,0,0,0"), ("3,1,2", "0,0,0")):
            pa, pb = plan(text_a, nl), plan(text_b, nl)
            assert [(lo, hi) for _, lo, hi in pa] == [(lo, hi) for _, lo, hi in pb]
            assert [dv for dv, _, _ in pa] == [int(t) for t in text_a.split(",")][:len(pa)]
            assert pa[0][1] == 0 and pa[-1][2] == nl and all(a[2] == b[1] for a, b in zip(pa, pa[1:])) and all(hi > lo for _, lo, hi in pa)
            assert max(hi - lo for _, lo, hi in pa) - min(hi - lo for _, lo, hi in pa) <= 1
    # ... and the program of a block is a function of (system, lambdas of the block, first circuit) only: same records and launches
    lams = [10 ** (-6 + 6 * k / 7) for k in range(8)]
    sysm = lgc.make_system(6, 64, 56, "cgd", 2, 0.0, 2, 1, 0, 0)
    for (_, lo, hi), (_, lo2, hi2) in zip(plan("0,1", 8), plan("0,0", 8)):
        pa = lgc.Program(sysm, lambdas=lams[lo:hi], first=lo)
        pb = lgc.Program(sysm, lambdas=lams[lo2:hi2], first=lo2)
        assert pa.records().tobytes() == pb.records().tobytes() and pa.launches() == pb.launches()


def test_bench_gpus_preflight_names_what_is_wrong():
    """bench.py --gpus N checks its devices before it builds the RCCL group (one GPU per rank, every pair reachable)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    assert "2 ranks but 1 visible GPU" in bench.devices_preflight(None, 2, 1)["error"]
    ok = bench.devices_preflight(None, 4, 8, can_access=lambda i, j: True)
    assert ok == {"distinct_devices": True, "peer_access": True, "no_peer_path": None}
    part = bench.devices_preflight(None, 3, 4, can_access=lambda i, j: (i, j) != (0, 2))
    assert part["peer_access"] is False and part["no_peer_path"] == ["0->2"]
