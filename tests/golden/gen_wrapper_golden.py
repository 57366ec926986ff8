#!/usr/bin/env python3
"""Golden vectors for the Python wrapper, produced by importing the REFERENCE wrapper
(/root/reference/python_interface/MPCLinearRegression.py) in the build container.

    python tests/golden/gen_wrapper_golden.py     # rewrites tests/golden/wrapper_golden.json

The reference module never travels: only its inputs/outputs are committed.  sklearn's
`linear_model.base` no longer exists, so the module is imported with a one-line alias
(SURVEY.md 8(c)).  The peer exchange (msgpack over TCP) is bypassed by pre-setting
`other_parameters`."""
import json
import os
import sys
import tempfile

import sklearn.linear_model._base as _base

sys.modules["sklearn.linear_model.base"] = _base
sys.path.insert(0, "/root/reference/python_interface")
import MPCLinearRegression as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CSV = """age;sex;height;weight;income
23;m;1.80;80.5;3000
31;w;1.65;60.0;4200
45;m;1.75;90.25;5100.5
52;w;1.60;55.75;3900
38;m;1.90;101.0;6100
29;w;1.70;64.5;2800.25
"""


def main():
    out = {}
    cols = [[1.0, 2.0, 3.0, 4.0], [0.5, -1.25, 7.75, 3.0, 3.0], [10.0, 10.5, 9.5], [1e6, 2e6, 4e6, 8e6, 1.6e7]]
    out["studentize"] = [dict(inp=c, out=list(ref.studentize(list(c)))) for c in cols]
    m = [list(c) for c in cols[:2]]
    sm = ref.studentize_matrix([list(r) for r in m])
    out["studentize_matrix"] = dict(inp=m, out=[sm[0], sm[1], sm[2]])

    fd, csv_path = tempfile.mkstemp(suffix=".csv")
    os.write(fd, CSV.encode()); os.close(fd)
    out["csv"] = CSV
    cases = []
    for spec, other in (("0 c1 2", dict(length=2, is_last=True)), ("3 r4", dict(length=4, is_last=False)),
                        ("2 0", dict(length=1, is_last=True)), ("c1 r4", dict(length=2, is_last=False))):
        r = ref.MPCLinearRegression("127.0.0.1:4000", "127.0.0.1:5000")
        r.exchange_parameters = lambda self=r, other=other: setattr(self, "other_parameters", dict(other))
        matrix = r.make_matrix(csv_path, spec)
        params = json.loads(json.dumps(r.parameters))
        path = r.make_csv(matrix)
        text = open(path).read()
        os.remove(path)
        cases.append(dict(spec=spec, other=other, matrix=matrix, parameters=params, csp_ip=r.csp_ip, eval_ip=r.eval_ip,
                          mpc_file=text))
    out["make"] = cases

    # predict: two sides' parameters combined; coefficients fixed
    a = ref.MPCLinearRegression("127.0.0.1:4000", "127.0.0.1:5000")
    a.exchange_parameters = lambda: setattr(a, "other_parameters", {"length": 2})
    a.make_matrix(csv_path, "0 c1 2")
    b = ref.MPCLinearRegression("127.0.0.1:5000", "127.0.0.1:4000")
    b.exchange_parameters = lambda: setattr(b, "other_parameters", {"length": 3})
    b.make_matrix(csv_path, "3 r4")
    a.other_parameters = json.loads(json.dumps(b.parameters))
    b.other_parameters = json.loads(json.dumps(a.parameters))
    coef = [0.25, -0.5, 0.125, 0.75]
    a.result = list(coef); b.result = list(coef)
    preds = []
    for X in ({"age": 40, "sex": "w", "height": 1.7, "weight": 70.0}, [35, "m", 1.82, 77.5],
              {"age": float("nan"), "sex": "m", "height": 1.7, "weight": float("nan")}, ["NaN", "w", 1.5, 50]):
        Xj = json.loads(json.dumps(X).replace("NaN", '"NaN"')) if False else X
        preds.append(dict(X=[("NaN" if (isinstance(v, float) and v != v) else v) for v in (X.values() if isinstance(X, dict) else X)],
                          keys=list(X.keys()) if isinstance(X, dict) else None,
                          a=a.predict(X), b=b.predict(X)))
    out["predict"] = dict(coef=coef, params_a=json.loads(json.dumps(a.parameters)), params_b=json.loads(json.dumps(b.parameters)),
                          cases=preds)
    os.remove(csv_path)

    import re
    line = "Result:    0.984331027786964    0.792399824970372   -0.754117840176144    0.592849130685193    0.057351715952213 "
    out["result_line"] = dict(line=line, parsed=[float(x) for x in re.findall("-?[0-9]+.[0-9]+", line)])
    json.dump(out, open(os.path.join(HERE, "wrapper_golden.json"), "w"), indent=1)
    print("wrapper golden vectors written")


if __name__ == "__main__":
    main()
