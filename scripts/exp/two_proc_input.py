"""the input file of bin/test_linear_system for a synthetic d x d system (what bench.py's two_process_ring writes)
   python scripts/exp/two_proc_input.py <d> <path>"""
import sys
import numpy as np
d, path = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(1000)
n = 4 * d
X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
y = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
Af = X.T @ X / (n * d) + np.eye(d) * 1e-3
bf = X.T @ y / (n * d)
with open(path, "w") as f:
    f.write("%d %d\n" % (d, d)); np.savetxt(f, Af, fmt="%.17g")
    f.write("%d\n" % d); np.savetxt(f, bf[None, :], fmt="%.17g")
    f.write("%d\n" % d); np.savetxt(f, np.zeros((1, d)), fmt="%g")
