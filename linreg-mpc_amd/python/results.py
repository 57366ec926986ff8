"""Result files in the layout of the reference's experiment drivers.

`parse_exec` reads the stdout of `bin/linreg` / `bin/test_linear_system` (party 2) the way
experiments/test_phase2_aws.py:70-141 does; `write_phase2_out` writes the `.out` file of
test_phase2_aws.py:143-186 (header row, per-iteration cgd rows, solution / objective / result /
condition number blocks), so that files produced with this build can be dropped next to
experiments/results/phase2_{32,64}/*.out and read by the same plotting code.
`write_phase1_out` writes the four-line JSON-ish layout of experiments/results/phase1/*.out.
"""
import re

import numpy as np


def objective(X, y, beta, lam, n):
    """experiments/generate_tests.py objective: 1/n ||X b - y||^2 + lambda ||b||^2."""
    X = np.asarray(X, dtype=float)
    y = np.asarray(y, dtype=float)
    beta = np.asarray(beta, dtype=float)
    r = X.dot(beta) - y
    return float(r.dot(r) / n + lam * beta.dot(beta))


_NUMS = re.compile(r"((\s*[\d\.-]+)+)\s*$")


def parse_exec(text, alg):
    """Returns dict(ot_time, time, gate_count, result, iter_solutions, iter_times, iter_gates)."""
    out = dict(ot_time=0.0, time=None, gate_count=None, ref_gate_count=None, result=None, iter_solutions=[], iter_times=[],
               iter_gates=[])
    for line in text.splitlines():
        m = re.match(r"OT\s+time:\s*(\S+)", line)
        if m:
            out["ot_time"] = float(m.group(1))
        if alg == "cgd":
            m = _NUMS.match(line)
            if m and line.strip():
                out["iter_solutions"].append([float(v) for v in line.split()])
            m = re.match(r"Iteration\s+([0-9]+)\s+time:\s*(.+)$", line)
            if m:
                assert int(m.group(1)) == len(out["iter_times"])
                out["iter_times"].append(float(m.group(2)))
            m = re.match(r"Iteration\s+([0-9]+)\s+gate\s+count:\s+(.+)$", line)
            if m:
                assert int(m.group(1)) == len(out["iter_gates"])
                out["iter_gates"].append(int(m.group(2)))
        m = re.match(r"Time\s+elapsed:\s*(\S+)", line)
        if m:
            out["time"] = float(m.group(1))
        m = re.match(r"Number\s+of\s+gates:\s*(\S+)", line)
        if m:
            out["gate_count"] = int(m.group(1))
        m = re.match(r"Reference-equivalent\s+gates:\s*(\S+)", line)
        if m:
            out["ref_gate_count"] = int(m.group(1))
        m = re.match(r"Result:\s*(.+)", line)
        if m:
            out["result"] = [float(v) for v in m.group(1).split()]
    return out


def write_phase2_out(path, n, d, alg, run, solution, X=None, y=None, lam=0.0, condition_number=float("nan"),
                     objective_value=None):
    """`run` is the dict returned by parse_exec (or built from Solver.iterations()/trace())."""
    solution = np.asarray(solution, dtype=float)
    result = np.asarray(run["result"], dtype=float)
    error = float(np.linalg.norm(result - solution))
    if objective_value is None:
        objective_value = objective(X, y, solution, lam, n) if X is not None else float("nan")
    # gate_count is THIS build's circuit; ref_gate_count (extra last column, -1 where unknown) is the reference's count for
    # the same solve, the figure experiments/results/phase2_{32,64}/*.out hold: rates computed from either stay comparable
    # (the column is added only when the run reported the figure: files without it are byte-for-byte the reference's layout)
    ref = run.get("ref_gate_count")
    lines = ["n d algorithm ot_time time error gate_count" + (" ref_gate_count" if ref is not None else ""),
             "{0} {1} {2} {3} {4} {5} {6}".format(n, d, alg, run["ot_time"], run["time"], error, run["gate_count"])
             + (" %d" % ref if ref is not None else "")]
    if alg == "cgd":
        gates = list(run["iter_gates"])
        # the reference shifts the per-iteration counts so that the last row equals the total
        # (test_phase2_aws.py:149-156)
        after = run["gate_count"] - gates[-1] if gates else 0
        gates = [g + after for g in gates]
        lines.append("iter_i error_i obj_i time_i gate_count_i")
        for i, sol in enumerate(run["iter_solutions"]):
            sol = np.asarray(sol, dtype=float)
            obj = objective(X, y, sol, lam, n) if X is not None else float("nan")
            lines.append("{0} {1} {2} {3} {4}".format(i + 1, float(np.linalg.norm(sol - solution)), obj,
                                                    run["iter_times"][i], gates[i] if gates else -1))
    lines += ["solution:", str(d), " ".join(repr(float(v)) for v in solution),
              "Objective function on solution:", str(objective_value),
              "result:", str(d), " ".join(repr(float(v)) for v in result),
              "Condition number:", str(condition_number)]
    with open(path, "w") as f:
        f.write("\n".join(lines))
    return error


def write_phase1_out(path, n, d, p, party, cputime, wait_time, realtime, sent=(), flushes=()):
    """experiments/results/phase1/*.out (secure_multiplication.c:98-104 prints the second line)."""
    with open(path, "w") as f:
        f.write('{{"n":"{0}", "d":"{1}", "p":"{2}"}}\n'.format(n, d, p))
        f.write('{{"party":"{0}", "cputime":"{1:f}", "wait_time":{2:f}, "realtime":"{3:f}"}}\n'.format(
            party, cputime, wait_time, realtime))
        f.write("[" + ", ".join(str(int(v)) for v in sent) + "]\n")
        f.write("[" + ", ".join(str(int(v)) for v in flushes) + "]\n")
