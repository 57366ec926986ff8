"""sklearn-style two-party wrapper around the `linreg` binary: counterpart of the reference's
python_interface/MPCLinearRegression.py (same constructor, `fit(csv, columns)`, `predict(X)`,
column-spec grammar "0 1 c2 r5", studentisation, peer exchange, input-file layout and result
parsing), so that a user of the reference can switch without changing calling code.

Flow of `fit` (reference file:line):
  make_matrix   parse the column spec (203-227), read the CSV (59-82), studentise each owned
                column with the population standard deviation (13-27), exchange parameters with
                the peer (126-138), zero-pad the peer's columns (152-159)
  make_csv      write the MPC input file for exactly two data providers, CSP / Evaluator endpoints
                at port + 10 (84-124)
  run_mpc       the side without the result column starts party 3 (DP1) and party 1 (CSP); the
                side with it starts party 4 (DP2) and party 2 (Evaluator), parses the LAST stdout
                line and sends the coefficients to the peer (161-200)
"""
import csv
import logging
import math
import os
import re
import subprocess
import tempfile

from msgpack_connection import create_connection

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_BINARY = os.path.join(os.path.dirname(_HERE), "host", "bin", "linreg")
_RESULT_NUMBER = re.compile("-?[0-9]+.[0-9]+")       # the reference's pattern (194), dot unescaped
_COLUMN_SPEC = re.compile("(r?c?)([0-9]+)")


def studentize(values):
    """(x - mean) / population-sigma for one column; returns (values, mean, sigma)"""
    n = len(values)
    mean = sum(values) / n
    sigma = math.sqrt(1.0 / n * sum([pow(x - mean, 2) for x in values]))
    return ([(x - mean) / sigma for x in values], mean, sigma)


def studentize_matrix(rows):
    means, sigmas = [], []
    for i, row in enumerate(rows):
        rows[i], m, s = studentize(row)
        means.append(m)
        sigmas.append(s)
    return (rows, means, sigmas)


def parse_result_line(line):
    return [float(x) for x in _RESULT_NUMBER.findall(line)]


def _shift_port(endpoint, delta):
    host, port = endpoint.split(":")
    return host + ":" + str(int(port) + delta)


class MPCLinearRegression:
    """Secure linear regression with one other party (semi-honest)."""

    def __init__(self, own_ip, other_ip, delimiter=";", category_mappings={"m": [1.0], "w": [0.0]},
                 mpc_binary_path=DEFAULT_BINARY, mpc_args=["56", "cholesky", "10", "0.001"], debug=False):
        self.own_ip, self.other_ip = own_ip, other_ip
        self.csp_ip = self.eval_ip = ""
        self.csv_file = ""
        self.delimiter = delimiter
        self.category_mappings = category_mappings
        self.mpc_binary_path = mpc_binary_path
        self.mpc_args = list(mpc_args)
        self.debug = debug
        self.parameters = {"is_last": False, "arith_means": [], "variances": [], "length": -1, "owned_columns": []}
        self.other_parameters = {}
        self.result = []
        logging.basicConfig(level=logging.DEBUG if debug else logging.WARNING)

    # ------------------------------------------------------------------ input side
    def parse_csv(self, csv_file):
        owned = self.parameters["owned_columns"]
        rows = []
        with open(csv_file, "r") as f:
            for lineno, record in enumerate(csv.reader(f, delimiter=self.delimiter)):
                out = []
                for j, (is_cat, index, name) in enumerate(owned):
                    cell = record[index]
                    if lineno == 0:                      # header: remember the column name
                        owned[j] = (is_cat, index, cell.strip())
                    elif is_cat > 0:
                        mapped = self.category_mappings[cell]
                        if self.parameters["is_last"] and len(mapped) > 1 and j == len(owned) - 1:
                            raise Exception("Result column can't be a category feature")
                        out += mapped
                        owned[j] = (len(mapped), index, name)
                    else:
                        out.append(float(cell))
                if out:
                    rows.append(out)
        return rows

    def exchange_parameters(self):
        host, port = (self.own_ip if self.parameters["is_last"] else self.other_ip).split(":")
        with create_connection(host, int(port) + 20, self.parameters["is_last"]) as link:
            link.write(self.parameters)
            self.other_parameters = link.read()

    def calculate_matrix(self):
        columns = [list(c) for c in zip(*self.parse_csv(self.csv_file))]
        columns, self.parameters["arith_means"], self.parameters["variances"] = studentize_matrix(columns)
        self.parameters["length"] = len(columns)
        self.exchange_parameters()
        pad = [[0.0] * len(columns[0])] * self.other_parameters["length"]
        columns = pad + columns if self.parameters["is_last"] else columns + pad
        return [list(r) for r in zip(*columns)]

    def make_matrix(self, csv_file, owned_columns):
        owned, result_col, is_last = [], None, False
        for token in owned_columns.split():
            m = _COLUMN_SPEC.search(token)
            kind, index = m.group(1), int(m.group(2))
            if kind == "c":
                owned.append((1, index, ""))
            elif kind == "rc":
                is_last, result_col = True, (1, index, "")
            elif kind == "r":
                is_last, result_col = True, (0, index, "")
            else:
                owned.append((0, int(m.group(0)), ""))
        if result_col is not None:                       # the result column always goes last
            owned.append(result_col)
        self.parameters["owned_columns"] = owned
        self.parameters["is_last"] = is_last
        self.csv_file = csv_file
        return self.calculate_matrix()

    def make_csv(self, matrix):
        first, second = (self.own_ip, self.other_ip) if not self.parameters["is_last"] else (self.other_ip, self.own_ip)
        self.csp_ip, self.eval_ip = _shift_port(first, 10), _shift_port(second, 10)
        split = self.parameters["length"] if not self.parameters["is_last"] else self.other_parameters["length"]
        n, m = len(matrix), len(matrix[0])
        fd, path = tempfile.mkstemp()
        with os.fdopen(fd, "w") as f:
            w = csv.writer(f, delimiter=" ")
            w.writerow([n, m - 1, 2])
            w.writerow([self.csp_ip])
            w.writerow([self.eval_ip])
            w.writerow([first, 0])
            w.writerow([second, split])
            w.writerow([n, m - 1])
            for row in matrix:
                w.writerow(row[0:-1])
            w.writerow([n])
            w.writerow([row[-1] for row in matrix])
        return path

    # ------------------------------------------------------------------ protocol side
    def run_mpc(self, path):
        def command(party):
            return [self.mpc_binary_path, path, self.mpc_args[0], str(party)] + self.mpc_args[1:]
        quiet = None if self.debug else subprocess.DEVNULL
        if not self.parameters["is_last"]:
            subprocess.Popen(command(3), stdout=quiet)           # DP1
            subprocess.Popen(command(1), stdout=quiet)           # CSP
            host, port = self.other_ip.split(":")
            with create_connection(host, int(port) + 20, False) as link:
                self.result = link.read()
        else:
            subprocess.Popen(command(4), stdout=quiet)           # DP2
            evaluator = subprocess.Popen(command(2), stdout=subprocess.PIPE)
            output, _ = evaluator.communicate()
            self.result = parse_result_line(output.splitlines()[-1].decode("UTF-8"))
            host, port = self.own_ip.split(":")
            with create_connection(host, int(port) + 20, True) as link:
                link.write(self.result)

    def fit(self, csv_file, owned_columns):
        path = self.make_csv(self.make_matrix(csv_file, owned_columns))
        if not os.path.exists(self.mpc_binary_path):
            raise Exception("Can't find mpc binary!")
        self.run_mpc(path)
        if not self.debug:
            os.remove(path)

    def predict(self, X):
        """X: dict keyed by CSV column names, or a list in CSV column order; NaN -> training mean"""
        if self.result == []:
            raise Exception("Please fit a model first!")
        mine, theirs = self.parameters, self.other_parameters
        a, b = (mine, theirs) if not mine["is_last"] else (theirs, mine)
        means = a["arith_means"] + b["arith_means"]
        sigmas = a["variances"] + b["variances"]
        columns = [tuple(c) for c in a["owned_columns"] + b["owned_columns"]]
        if len(means) != len(sigmas):
            raise Exception("Length of means != length of variances. This should not happen.")
        slots = [None] * len(columns)
        if isinstance(X, dict):
            for i, (is_cat, _, name) in enumerate(columns[:-1]):
                slots[i] = self.category_mappings[X[name]] if is_cat > 0 else float(X[name])
        elif isinstance(X, list):
            order = sorted(((i, index) for i, (_, index, _) in enumerate(columns[:-1])), key=lambda t: t[1])
            for x, (i, _) in zip(X, order):
                try:
                    slots[i] = float(x)
                except ValueError:
                    slots[i] = self.category_mappings[x]
        else:
            raise Exception("input must be a dict or a list")
        flat = []
        for v in slots:
            if v is None:
                continue
            flat += v if isinstance(v, list) else [v]
        flat = [means[i] if math.isnan(v) else v for i, v in enumerate(flat)]
        acc = 0
        for i, v in enumerate(flat):
            acc += ((v - means[i]) / sigmas[i]) * self.result[i]
        return acc * sigmas[-1] + means[-1]
