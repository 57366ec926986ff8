"""Two-process phase 2 (bin/test_linear_system, the reference's benchmark binary) with the garbled
tables over the socket vs. in the device-resident ring.  Usage: gpu_ring_compare.py [d] [iters] [socket,ring]"""
import os, re, socket, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "linreg-mpc_amd", "host", "bin", "test_linear_system")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rng = np.random.default_rng(5)
X = rng.standard_normal((4 * d, d)); X /= np.abs(X).max(axis=0)
A = X.T @ X / (4 * d * d) + np.eye(d) * 1e-2
sol = rng.random(d); b = A @ sol
path = "/tmp/ring_ls_%d.in" % d
with open(path, "w") as f:
    f.write("%d %d\n" % (d, d))
    for i in range(d):
        f.write(" ".join(repr(float(v)) for v in A[i]) + " \n")
    f.write("%d\n" % d + " ".join(repr(float(v)) for v in b) + " \n")
    f.write("%d\n" % d + " ".join(repr(float(v)) for v in sol) + " ")
results = {}
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["socket", "ring"]
for mode, opt in (("socket", []), ("ring", ["--table_ring=%s" % os.environ.get("RING_SLOTS", "8")])):
    if mode not in modes:
        continue
    port = 12000 + int.from_bytes(os.urandom(2), "little") % 18000     # below the ephemeral port range
    t0 = time.time()
    procs = [subprocess.Popen([EXE, str(port), str(k), path, "cgd", str(iters), "56", "--host=127.0.0.1"] + opt,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k in (1, 2)]
    outs = [p.communicate(timeout=3000) for p in procs]
    wall = time.time() - t0
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    ev = outs[1][0].decode()
    te = float(re.search(r"Time elapsed:\s*(\S+)", ev).group(1))
    ng = int(re.search(r"Number of gates:\s*(\S+)", ev).group(1))
    it = [float(v) for v in re.findall(r"Iteration \d+ time: (\S+)", ev)]
    res = ev.strip().splitlines()[-1]
    results[mode] = res
    print("d=%d cgd-%d %-6s: time elapsed %.3fs (wall %.2fs incl. process start), gates %.3e -> %.3e AND/s, last-first iter %.3fs"
          % (d, iters, mode, te, wall, ng, ng / te, it[-1] - it[0] if it else 0), flush=True)
if len(results) == 2:
    assert results["socket"] == results["ring"], "results differ"
    print("identical results:", results["ring"][:80], "...")
