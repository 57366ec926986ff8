#!/usr/bin/env python3
"""Tolerance pins taken from data files the REFERENCE holds (run in the build container only;
/root/reference never travels -- only numbers are committed):

  matlab_systems.json    experiments/matlab/phillipp4.data (5x5) and adria1.data (20x20): rows
                         1..d = A, row d+1 = b, row d+2 = the solution z the MATLAB study compares
                         against (experiments/matlab/test7.m:13-16)
  reference_errors.json  the `error` column (||result - solution||_2, experiments/test_phase2_aws.py:145)
                         and the per-iteration errors of every experiments/results/phase2_{32,64}/*.out:
                         what the reference's own 32/64-bit solvers achieved on the generate_tests
                         distribution (n = 1e5, sigma = 0.1, 20 CGD iterations, precisions 30 / 56)

    python tests/golden/gen_reference_pins.py
"""
import glob
import json
import os
import re

REF = "/root/reference/experiments"
HERE = os.path.dirname(os.path.abspath(__file__))


def matlab(name, d):
    rows = [list(map(float, l.split())) for l in open(os.path.join(REF, "matlab", name)) if l.strip()]
    assert len(rows) == d + 2 and all(len(r) == d for r in rows), (name, len(rows))
    return {"name": name, "d": d, "A": rows[:d], "b": rows[d], "z": rows[d + 1]}


def errors():
    out = []
    for folder in ("phase2_32", "phase2_64"):
        for path in sorted(glob.glob(os.path.join(REF, "results", folder, "*.out"))):
            m = re.search(r"test_LS_(\d+)x(\d+)_([0-9.]+)_(\d+)_(cgd|cholesky)_(32|64)_(\d+)_p2\.out$", path)
            if not m:
                continue
            lines = open(path).read().split("\n")
            n, d, alg, _ot, _time, err, gates = lines[1].split()
            assert int(n) == int(m.group(1)) and int(d) == int(m.group(2)) and alg == m.group(5)
            rec = {"file": folder + "/" + os.path.basename(path), "n": int(n), "d": int(d), "alg": alg, "width": int(m.group(6)),
                   "sigma": float(m.group(3)), "iters": int(m.group(7)), "error": float(err), "gate_count": int(gates)}
            if alg == "cgd":
                it, ig = [], []
                for l in lines[3:]:
                    f = l.split()
                    if len(f) != 5 or not f[0].isdigit():
                        break
                    it.append(float(f[1]))
                    ig.append(int(f[4]))
                rec["iter_errors"] = it
                rec["iter_gates"] = ig          # cumulative gate count after iteration i (column gate_count_i)
            out.append(rec)
    return out


if __name__ == "__main__":
    json.dump([matlab("phillipp4.data", 5), matlab("adria1.data", 20)], open(os.path.join(HERE, "matlab_systems.json"), "w"), indent=0)
    errs = errors()
    json.dump(errs, open(os.path.join(HERE, "reference_errors.json"), "w"), indent=0)
    print(len(errs), "result files")
