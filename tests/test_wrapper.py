"""Python wrapper counterpart vs golden vectors captured from the reference wrapper
(tests/golden/gen_wrapper_golden.py; the reference module itself never travels)."""
import json
import math
import os
import socket
import tempfile
import threading

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
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
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
