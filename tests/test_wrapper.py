"""Python wrapper counterpart vs golden vectors captured from the reference wrapper
(tests/golden/gen_wrapper_golden.py; the reference module itself never travels)."""
import json
import math
import os
import socket
import tempfile
import threading

import numpy as np

import pytest

import mpc_linear_regression as mlr
import msgpack_connection as mpk

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wrapper_golden.json")))


def test_studentize():
    for c in GOLD["studentize"]:
        vals, mean, sigma = mlr.studentize(list(c["inp"]))
        assert [vals, mean, sigma] == c["out"]
    g = GOLD["studentize_matrix"]
    out = mlr.studentize_matrix([list(r) for r in g["inp"]])
    assert [out[0], out[1], out[2]] == g["out"]


@pytest.mark.parametrize("case", GOLD["make"], ids=lambda c: c["spec"])
def test_make_matrix_and_input_file(case, tmp_path):
    csvf = tmp_path / "data.csv"
    csvf.write_text(GOLD["csv"])
    r = mlr.MPCLinearRegression("127.0.0.1:4000", "127.0.0.1:5000")
    r.exchange_parameters = lambda: setattr(r, "other_parameters", dict(case["other"]))
    matrix = r.make_matrix(str(csvf), case["spec"])
    assert matrix == case["matrix"]
    assert json.loads(json.dumps(r.parameters)) == case["parameters"]
    path = r.make_csv(matrix)
    text = open(path, newline="").read()
    os.remove(path)
    assert (r.csp_ip, r.eval_ip) == (case["csp_ip"], case["eval_ip"])
    assert text.replace("\r\n", "\n") == case["mpc_file"].replace("\r\n", "\n")


def test_predict_dict_list_and_nan():
    g = GOLD["predict"]
    for side, params, other in (("a", g["params_a"], g["params_b"]), ("b", g["params_b"], g["params_a"])):
        r = mlr.MPCLinearRegression("127.0.0.1:1", "127.0.0.1:2")
        r.parameters = json.loads(json.dumps(params)); r.other_parameters = json.loads(json.dumps(other))
        r.result = list(g["coef"])
        for c in g["cases"]:
            vals = [float("nan") if v == "NaN" and c["keys"] else v for v in c["X"]]
            X = dict(zip(c["keys"], vals)) if c["keys"] else list(c["X"])
            assert r.predict(X) == c[side]
    with pytest.raises(Exception):
        mlr.MPCLinearRegression("a:1", "b:2").predict([1])


def test_result_line_regex():
    g = GOLD["result_line"]
    assert mlr.parse_result_line(g["line"]) == g["parsed"]


def test_msgpack_peer_link_roundtrip():
    from helpers import free_ports
    port = free_ports(1)[0]
    got = {}

    def server():
        with mpk.create_connection("127.0.0.1", port, True) as c:
            got["srv"] = c.read()
            c.write({"is_last": True, "arith_means": [1.5], "owned_columns": [[0, 3, "x"]]})
    t = threading.Thread(target=server); t.start()
    with mpk.create_connection("127.0.0.1", port, False) as c:
        c.write({"length": 3, "variances": [0.25, 2.0]})
        got["cli"] = c.read()
    t.join(timeout=10)
    assert got["srv"] == {"length": 3, "variances": [0.25, 2.0]}
    assert got["cli"] == {"is_last": True, "arith_means": [1.5], "owned_columns": [[0, 3, "x"]]}


def test_result_file_layout(tmp_path):
    """.out layout of experiments/test_phase2_aws.py:143-186 from a party-2 stdout transcript"""
    import results
    text = "\n".join([
        "", "Algorithm: cgd", "OT time: 0.250000", "Starting iterations.",
        "Iteration 0 (x):", "   0.500000000000000    0.250000000000000 ",
        "Gamma:       0.10000000000000000000 ", "Eta:       0.20000000000000000000 ",
        "q:       0.30000000000000000000 ", "ng:       0.40000000000000000000 ",
        "Iteration 0 gate count: 1000", "Iteration 0 time: 0.100000",
        "Iteration 1 (x):", "   1.000000000000000   -2.000000000000000 ",
        "Gamma:       0.10000000000000000000 ", "Eta:       0.20000000000000000000 ",
        "q:       0.30000000000000000000 ", "ng:       0.40000000000000000000 ",
        "Iteration 1 gate count: 1900", "Iteration 1 time: 0.200000",
        "Time elapsed: 0.450000", "Number of gates: 2000",
        "Result:    1.000000000000000   -2.000000000000000 ", ""])
    run = results.parse_exec(text, "cgd")
    assert run["iter_solutions"] == [[0.5, 0.25], [1.0, -2.0]] and run["iter_gates"] == [1000, 1900]
    assert run["iter_times"] == [0.1, 0.2] and run["gate_count"] == 2000 and run["ot_time"] == 0.25
    X = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]); y = np.array([1.0, -2.0, -1.0])
    path = str(tmp_path / "t.out")
    err = results.write_phase2_out(path, 3, 2, "cgd", run, [1.0, -2.0], X=X, y=y, lam=0.5, condition_number=3.0)
    rows = open(path).read().split("\n")
    assert err == 0.0
    assert rows[0] == "n d algorithm ot_time time error gate_count"
    assert rows[1] == "3 2 cgd 0.25 0.45 0.0 2000"
    assert rows[2] == "iter_i error_i obj_i time_i gate_count_i"
    assert rows[3].split()[0] == "1" and rows[3].split()[4] == "1100" and rows[4].split()[4] == "2000"
    assert float(rows[4].split()[1]) == 0.0 and abs(float(rows[4].split()[2]) - 2.5) < 1e-12
    assert rows[5:] == ["solution:", "2", "1.0 -2.0", "Objective function on solution:", "2.5",
                        "result:", "2", "1.0 -2.0", "Condition number:", "3.0"]
    results.write_phase1_out(str(tmp_path / "p.out"), 10, 5, 3, 4, 1.5, 0.25, 2.0, sent=[0, 10], flushes=[0, 2])
    rows = open(str(tmp_path / "p.out")).read().splitlines()
    assert rows[0] == '{"n":"10", "d":"5", "p":"3"}'
    assert rows[1] == '{"party":"4", "cputime":"1.500000", "wait_time":0.250000, "realtime":"2.000000"}'
    assert rows[2:] == ["[0, 10]", "[0, 2]"]


def _fit_side(own, other, csv_path, spec, args, q):
    """one MPCLinearRegression.fit() in its own process; keeps the MPC input file it wrote"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "linreg-mpc_amd", "python"))
    import mpc_linear_regression as m
    r = m.MPCLinearRegression(own, other, mpc_args=args)
    kept = {}
    make_csv = r.make_csv
    def keep(matrix):
        path = make_csv(matrix)
        kept["text"] = open(path).read()
        return path
    r.make_csv = keep
    r.fit(csv_path, spec)
    q.put((spec, r.result, kept["text"], r.predict({"age": 40.0, "sex": "m", "height": 1.82, "weight": 77.0})))


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["cgd", "cholesky"])
def test_fit_two_instances_end_to_end(tmp_path, oracle, alg):
    """python_interface/MPCLinearRegression.py:165-194, 229-244: two wrapper instances on localhost
    exchange parameters over msgpack, write their input files, spawn DP1 + CSP / DP2 + Evaluator
    (bin/linreg on the GPU), the evaluator side parses the last stdout line and hands the
    coefficients to the peer.  The coefficients equal the oracle's on the combined data set."""
    import multiprocessing as mp
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "linreg-mpc_amd", "host")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(3)
    n = 60
    age = rng.integers(20, 70, n).astype(float); sex = rng.integers(0, 2, n)
    height = 1.5 + 0.4 * rng.random(n); weight = 50 + 40 * rng.random(n)
    income = 800 + 35 * age + 400 * sex + 900 * height - 3 * weight + 50 * rng.standard_normal(n)
    csvf = tmp_path / "people.csv"
    with open(csvf, "w") as f:
        f.write("age;sex;height;weight;income\n")
        for i in range(n):
            f.write("%r;%s;%r;%r;%r\n" % (float(age[i]), "mw"[1 - int(sex[i])], float(height[i]), float(weight[i]), float(income[i])))
    from helpers import free_ports
    base = free_ports(1)[0]          # (below the ephemeral range; base + 100 likewise)
    a_ip, b_ip = "127.0.0.1:%d" % base, "127.0.0.1:%d" % (base + 100)
    args = ["56", alg, "12", "0.001"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pa = ctx.Process(target=_fit_side, args=(a_ip, b_ip, str(csvf), "0 c1", args, q))          # age, sex
    pb = ctx.Process(target=_fit_side, args=(b_ip, a_ip, str(csvf), "2 3 r4", args, q))        # height, weight, income
    pa.start(); pb.start()
    try:
        outs = dict((o[0], o[1:]) for o in (q.get(timeout=120), q.get(timeout=120)))
    finally:
        pa.join(20); pb.join(20)
        for pr in (pa, pb):
            if pr.is_alive():
                pr.kill()
    assert pa.exitcode == 0 and pb.exitcode == 0
    res_a, file_a, pred_a = outs["0 c1"]
    res_b, file_b, pred_b = outs["2 3 r4"]
    assert res_a == res_b and len(res_b) == 4                     # the peer received what the evaluator side parsed
    assert pred_a == pred_b
    # the data set the four processes computed on: columns 0..1 from side a's file, 2..3 and y from side b's
    ta, tb = file_a.split("\n"), file_b.split("\n")
    assert ta[:6] == tb[:6]                                       # same header: n d P, CSP, Evaluator, the two providers
    rows = [ra.split()[:2] + rb.split()[2:] for ra, rb in zip(ta[6:6 + n], tb[6:6 + n])]
    comb = tmp_path / "combined.in"
    comb.write_text("\n".join(ta[:6] + [" ".join(r) for r in rows] + tb[6 + n:]))
    beta = oracle.linreg_file(str(comb), 56, -1, 64, 64, {"cholesky": 0, "cgd": 2}[alg], 12, 0.001)
    assert res_b == [float("%.15f" % (int(v) / 2.0 ** 56)) for v in beta]
    # and it is a sensible regression: close to least squares on the studentised data
    X = np.array([[float(v) for v in r] for r in rows]); y = np.array([float(v) for v in tb[6 + n + 1].split()])
    ls = np.linalg.solve(X.T @ X / (n * 4) + 0.001 * np.eye(4), X.T @ y / (n * 4))
    assert np.allclose(res_b, ls, atol=5e-3 if alg == "cgd" else 1e-6)
