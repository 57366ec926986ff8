"""two-process d=500 CGD-15 through the hipIpc table ring with different slot counts"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
d, iters, p = 500, 15, 56
rng = np.random.default_rng(1000)
n = 4 * d
X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
y = X @ rng.random(d) + 0.1 * rng.standard_normal(n)
Af = X.T @ X / (n * d) + np.eye(d) * 1e-3
bf = X.T @ y / (n * d)
import subprocess, re, tempfile, time
exe = os.path.join(ROOT, "linreg-mpc_amd", "host", "bin", "test_linear_system")
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "ls.in")
with open(path, "w") as f:
    f.write("%d %d\n" % (d, d)); np.savetxt(f, Af, fmt="%.17g"); f.write("%d\n" % d); np.savetxt(f, bf[None, :], fmt="%.17g")
    f.write("%d\n" % d); np.savetxt(f, np.zeros((1, d)), fmt="%g")
for slots in sys.argv[1:]:
    port = bench._free_ports(1)[0]
    t0 = time.time()
    procs = [subprocess.Popen([exe, str(port), str(k), path, "cgd", str(iters), str(p), "--host=127.0.0.1", "--table_ring=" + slots],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in (1, 2)]
    outs = [q.communicate(timeout=900) for q in procs]
    ev = outs[1][0].decode()
    its = [float(v) for v in re.findall("Iteration [0-9]+ time: ([0-9.]+)", ev)]
    print("slots", slots, "rc", [q.returncode for q in procs], "last iteration time", its[-1] if its else None, "wall %.2f" % (time.time() - t0), outs[0][1].decode()[-200:], flush=True)
