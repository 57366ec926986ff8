"""BASELINE.json configs 3, 4 and 5 at full size, exact against the oracle, plus the large direct
solve -- every party a separate process where the config says so (bin/linreg on this box, one GPU).

  C3  n = 1e4, d = 100, 2 data providers, CGD-15, 64-bit, --use_ot phase 1        (Result line == oracle)
  C4  n = 5e4, d = 500, 5 data providers, CGD-20, phase 1 in 64 bits / phase 2 in 32 bits
      (--prec_phase2=30), TI mode with a pinned TI seed: exact at SHARE level -- every share is shifted
      on its own (src/phase1.c:609-638), so the oracle replays the TI's AES-CTR stream pair by pair
      (1e5 cross pairs x (2n+1) words, ~35 s of host time; the protocol run itself takes ~22 s)
  C5  all 64 lambdas of the d = 100 CGD-15 circuit as one merged program, every beta == oracle
  Cholesky d = 500, 64-bit (1.9e11 AND gates)
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import oracle_solve, split_shares, synth_system
from test_host import HOST, _free_ports, _run_all

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))


def _write_instance(path, n, d, starts, seed):
    """experiments/generate_tests.py:159-169 distribution in the README.md:51-77 file format"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    beta = rng.random(d)
    y = X @ beta + 0.1 * rng.standard_normal(n)
    ports = _free_ports(len(starts) + 2)
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (n, d, len(starts)))
        f.write("127.0.0.1:%d\n127.0.0.1:%d\n" % (ports[0], ports[1]))
        for k, st in enumerate(starts):
            f.write("127.0.0.1:%d %d\n" % (ports[2 + k], st))
        f.write("%d %d\n" % (n, d))
        np.savetxt(f, X, fmt="%.17g")
        f.write("%d\n" % n)
        np.savetxt(f, y[None, :], fmt="%.17g")
    return X, y


@pytest.mark.parametrize("ot", ["--use_ot", "--ot_ring", "--ot_ring --input_ring"])
def test_config3_use_ot_full_size(tmp_path, oracle, ot):
    """bin/linreg <file> 56 <party> cgd 15 0.001 --use_ot --table_ring with n = 1e4, d = 100, P = 2
    (--ot_ring: the 39 GB of OT-extension messages between the two providers stay in HBM)"""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    path = str(tmp_path / "c3.in")
    _write_instance(path, 10000, 100, [0, 50], 3)
    outs = _run_all(path, 2, ["56", "cgd", "15", "0.001"] + ot.split() + ["--table_ring"], timeout=900)
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    beta = oracle.linreg_file(path, 56, -1, 64, 64, 2, 15, 0.001)
    assert got == ["%.15f" % (int(v) / 2.0 ** 56) for v in beta]
    assert "Number of gates:" in outs[1]


@pytest.mark.parametrize("ring", [[], ["--ti_ring"], ["--ti_ring", "--input_ring"]], ids=["sockets", "ti-ring", "ti-and-input-ring"])
def test_config4_five_providers_64_32_split_share_level(tmp_path, oracle, gccpu, ring):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    n, d, starts = 50000, 500, [0, 100, 200, 300, 400]
    p1, p2, lam, iters = 56, 30, 0.001, 20
    path = str(tmp_path / "c4.in")
    _write_instance(path, n, d, starts, 4)
    seed = bytes(range(0x60, 0x70))
    os.environ["LINREG_TI_SEED"] = seed.hex()
    try:
        outs = _run_all(path, 5, [str(p1), "cgd", str(iters), str(lam), "--width_phase2=32", "--prec_phase2=%d" % p2, "--table_ring"] + ring,
                        timeout=1500, exe_name="linreg_testhooks")
    finally:
        del os.environ["LINREG_TI_SEED"]
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1].strip().splitlines()[-1])
    inp = oracle.read_input(path)
    assert (inp["n"], inp["d"], inp["start"]) == (n, d, starts)
    Xq = oracle.quantize(inp["X"], p1, n, 32); yq = oracle.quantize(inp["y"], p1, n, 32)   # cast through the 32-bit fixed_t
    sA, sb, used = oracle.ti_shares_stream(Xq.reshape(n, d), yq, n, d, p1, 64, starts,
                                           lambda first, count: gccpu.ti_stream_words(seed, first, count, 64))
    assert used == 100400 * (2 * n + 1)                      # cross-party pairs of this column split
    cA = np.stack([oracle.convert_shares(r, p1, p2, 64, 32) for r in sA])
    cb = np.stack([oracle.convert_shares(r, p1, p2, 64, 32) for r in sb])
    a, bb = oracle.circuit_input(oracle.sum_shares(cA, 32), oracle.sum_shares(cb, 32), d, lam, p2, 32)
    beta = oracle.cgd(a, bb, d, p2, 32, iters)
    assert got == ["%.15f" % (int(v) / 2.0 ** p2) for v in beta]


def test_config5_all_64_lambdas(lgc, oracle):
    import sweep
    rng = np.random.default_rng(0)
    d, n, P, w, p, iters = 100, 10000, 2, 64, 56, 15
    A, b = synth_system(oracle, rng, n, d, w, p)
    sh = split_shares(rng, A, b, P, w)
    lams = sweep.c5_lambdas(64)
    sv = lgc.Solver(lgc.make_system(d, w, p, "cgd", iters, 0.0, P, 1, 0, 0), seed=bytes(range(16)), lambdas=lams)
    sv.set_shares(sh)
    sv.run()
    res = sv.beta()
    sv.close()
    for t, lam in enumerate(lams):
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cgd", iters, lam, 1)
        assert [int(v) for v in res[t]] == [int(v) for v in exp], (t, lam)


def test_cholesky_d500_bit_exact(lgc, oracle):
    rng = np.random.default_rng(8)
    d, n, w, p = 500, 1500, 64, 56
    A, b = synth_system(oracle, rng, n, d, w, p)
    sh = split_shares(rng, A, b, 2, w)
    sv = lgc.Solver(lgc.make_system(d, w, p, "cholesky", 0, 0.0, 2, 0, 0, 0), seed=bytes(range(16)))
    sv.set_shares(sh)
    sv.run()
    exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cholesky", 0, 0.0, 0)
    assert sv.beta().tolist() == exp.tolist()
    assert sv.stats()["and_gates"] > 1e11
    sv.close()
