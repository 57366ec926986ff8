"""config 2 through bin/linreg with LINREG_TRACE=1: where the wall clock of a small end-to-end run goes"""
import sys, os, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
n, d, starts = 1000, 20, [0, 10]
rng = np.random.default_rng(7)
X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
y = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
exe = os.path.join(ROOT, "linreg-mpc_amd", "host", "bin", "linreg")
for rep in range(3):
    ports = bench._free_ports(4)
    path = "/tmp/c2t.in"
    with open(path, "w") as f:
        f.write("%d %d %d\n127.0.0.1:%d\n127.0.0.1:%d\n" % (n, d, 2, ports[0], ports[1]))
        for k, st in enumerate(starts): f.write("127.0.0.1:%d %d\n" % (ports[2 + k], st))
        f.write("%d %d\n" % (n, d)); np.savetxt(f, X, fmt="%.17g"); f.write("%d\n" % n); np.savetxt(f, y[None, :], fmt="%.17g")
    env = dict(os.environ, LINREG_TRACE="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([exe, path, "56", str(k), "cholesky", "0", "0.001", "--table_ring"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for k in range(1, 5)]
    outs = [q.communicate(timeout=120) for q in procs]
    print("rep", rep, "wall %.3f" % (time.perf_counter() - t0), flush=True)
    if rep == 2:
        for k in (0, 1):
            print("".join(l + "\n" for l in outs[k][1].decode().splitlines() if l.startswith("[party")))
