"""BASELINE.json configs 2-4 end to end through bin/linreg (all parties on this box, one GPU):
generates the synthetic instance (experiments/generate_tests.py:159-169 distribution), runs the
processes, checks party 2's Result line against the oracle, reports wall-clock."""
import sys, os, re, time, socket, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, gccpu
oracle = orc.load()
mirror = gccpu.load()
TI_SEED = bytes(range(0x60, 0x70))
HOST = os.path.join(ROOT, "linreg-mpc_amd", "host")

from helpers import free_ports

def write_instance(path, n, d, starts, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0)
    beta = rng.random(d); y = X @ beta + 0.1 * rng.standard_normal(n)
    ports = free_ports(len(starts) + 2)
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (n, d, len(starts)))
        f.write("127.0.0.1:%d\n127.0.0.1:%d\n" % (ports[0], ports[1]))
        for k, st in enumerate(starts): f.write("127.0.0.1:%d %d\n" % (ports[2 + k], st))
        f.write("%d %d\n" % (n, d))
        for i in range(n): f.write(" ".join(repr(float(v)) for v in X[i]) + "\n")
        f.write("%d\n" % n + " ".join(repr(float(v)) for v in y) + "\n")
    return X, y

def run(name, n, d, starts, alg, iters, lam, extra, w2=64, p1=56, p2=None, seed=0):
    path = "/tmp/%s.in" % name
    t0 = time.time(); X, y = write_instance(path, n, d, starts, seed); t1 = time.time()
    exe = os.path.join(HOST, "bin", "linreg_testhooks" if w2 == 32 else "linreg")   # 64->32: the TI seed is pinned (below)
    P = len(starts)
    args = [str(p1), alg, str(iters), repr(lam)] + extra
    t2 = time.time()
    env = dict(os.environ, LINREG_TI_SEED=TI_SEED.hex()) if w2 == 32 else dict(os.environ)
    procs = [subprocess.Popen([exe, path, args[0], str(k)] + args[1:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for k in range(1, P + 3)]
    outs = [p.communicate(timeout=3000) for p in procs]
    t3 = time.time()
    for p, (o, e) in zip(procs, outs):
        if p.returncode != 0: print("FAILED", e.decode()[-500:]); return
    ev = outs[1][0].decode()
    if os.environ.get("CONFIG_RUNS_DUMP"):
        os.makedirs(os.environ["CONFIG_RUNS_DUMP"], exist_ok=True)
        for k, (o, e) in enumerate(outs):
            keep = [l for l in o.decode().splitlines() if len(l) < 300 and re.search("time|Time|finished|OT|party", l)]
            open(os.path.join(os.environ["CONFIG_RUNS_DUMP"], "%s_party%d.txt" % (name, k + 1)), "w").write("\n".join(keep[-60:]) + "\n--- stderr\n" + e.decode()[-3000:])
    got = re.findall("-?[0-9]+\\.[0-9]+", ev.strip().splitlines()[-1])
    elapsed = float(re.search("Time elapsed: ([0-9.]+)", ev).group(1))
    gates = int(re.search("Number of gates: ([0-9]+)", ev).group(1))
    exact = None
    if w2 == 64:
        inp = oracle.read_input(path)
        beta = oracle.linreg_file(path, p1, -1, 64, 64, {"cholesky": 0, "ldlt": 1, "cgd": 2}[alg], iters, lam)
        exact = got == ["%.15f" % (int(v) / 2.0 ** p1) for v in beta]
    elif os.environ.get("CONFIG_RUNS_EXACT32", "1") != "0":
        # 64 -> 32 bit: every share is shifted on its own (src/phase1.c:609-638), so the oracle replays the pinned
        # TI stream pair by pair and converts share by share (as tests/test_gpu_configs.py does at n = 5 000)
        import numpy as np
        t_or = time.time()
        inp = oracle.read_input(path)
        Xq = oracle.quantize(inp["X"], p1, n, 32); yq = oracle.quantize(inp["y"], p1, n, 32)
        sA, sb, used = oracle.ti_shares_stream(Xq.reshape(n, d), yq, n, d, p1, 64, starts,
                                               lambda first, count: mirror.ti_stream_words(TI_SEED, first, count, 64))
        p2v = [int(a.split("=")[1]) for a in extra if a.startswith("--prec_phase2=")][0]
        cA = np.stack([oracle.convert_shares(r, p1, p2v, 64, 32) for r in sA])
        cb = np.stack([oracle.convert_shares(r, p1, p2v, 64, 32) for r in sb])
        a, bb = oracle.circuit_input(oracle.sum_shares(cA, 32), oracle.sum_shares(cb, 32), d, lam, p2v, 32)
        beta = oracle.cgd(a, bb, d, p2v, 32, iters)
        exact = got == ["%.15f" % (int(v) / 2.0 ** p2v) for v in beta]
        print("oracle replay of the TI stream: %.0f s" % (time.time() - t_or), file=sys.stderr)
    # distance to the double-precision ridge solution of the same (normalised) system
    import numpy as np
    A = X.T @ X / (n * d); A[np.diag_indices(d)] += lam
    ref = np.linalg.solve(A, X.T @ y / (n * d))
    err = float(np.max(np.abs(np.array([float(v) for v in got]) - ref)))
    ot = re.search("Time taken for OT: ([0-9.]+)", ev)
    its = [float(v) for v in re.findall("Iteration [0-9]+ time: ([0-9.]+)", ev)]
    print(json.dumps(dict(config=name, n=n, d=d, P=P, alg=alg, iters=iters, opts=extra, wall_all_processes_s=round(t3 - t2, 2),
                          evaluator_time_elapsed_s=elapsed, evaluator_inputs_received_s=float(ot.group(1)) if ot else None,
                          last_iteration_s=its[-1] if its else None, input_file_write_s=round(t1 - t0, 1),
                          and_gates=gates, exact_vs_oracle=exact, max_abs_err_vs_float_solve=err)), flush=True)

which = sys.argv[1:] or ["c2", "c3"]
# the same runs over gate hash 1 (bin/linreg --gate_hash=chaskey12): names get the suffix "-chaskey12"
HASH = ["--gate_hash=chaskey12"]
if "c2rh" in which: run("c2-ring-chaskey12", 1000, 20, [0, 10], "cholesky", 0, 0.001, ["--table_ring"] + HASH)
if "c3titrh" in which: run("c3ti-ti-ring-chaskey12", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--table_ring", "--ti_ring"] + HASH)
if "c4trh" in which: run("c4-ti-ring-chaskey12", 50000, 500, [0, 100, 200, 300, 400], "cgd", 20, 0.001,
                         ["--width_phase2=32", "--prec_phase2=30", "--table_ring", "--ti_ring"] + HASH, w2=32)
subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
if "c2" in which: run("c2", 1000, 20, [0, 10], "cholesky", 0, 0.001, [])
if "c3ti" in which: run("c3ti", 10000, 100, [0, 50], "cgd", 15, 0.001, [])
if "c3" in which: run("c3", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--use_ot"])
if "c3l" in which: run("c3ti-lanes4", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--table_lanes=4"])
if "c3l8" in which: run("c3ti-lanes8", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--table_lanes=8"])
if "c3or" in which: run("c3-otring", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--ot_ring", "--table_ring"])
if "c4" in which: run("c4", 50000, 500, [0, 100, 200, 300, 400], "cgd", 20, 0.001, ["--width_phase2=32", "--prec_phase2=30"], w2=32)
# --table_ring: CSP and Evaluator processes share the garbled tables in HBM (hipIpc) instead of the socket
if "c2r" in which: run("c2-ring", 1000, 20, [0, 10], "cholesky", 0, 0.001, ["--table_ring"])
if "c3tir" in which: run("c3ti-ring", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--table_ring"])
if "c3r" in which: run("c3-ring", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--use_ot", "--table_ring"])
if "c4s" in which: run("c4-small-n-ring", 1000, 500, [0, 100, 200, 300, 400], "cgd", 20, 0.001,
                       ["--width_phase2=32", "--prec_phase2=30", "--table_ring"], w2=32)
if "c4r" in which: run("c4-ring", 50000, 500, [0, 100, 200, 300, 400], "cgd", 20, 0.001,
                       ["--width_phase2=32", "--prec_phase2=30", "--table_ring"], w2=32)
if "c4tr" in which: run("c4-ti-ring", 50000, 500, [0, 100, 200, 300, 400], "cgd", 20, 0.001,
                        ["--width_phase2=32", "--prec_phase2=30", "--table_ring", "--ti_ring"], w2=32)
if "c3titr" in which: run("c3ti-ti-ring", 10000, 100, [0, 50], "cgd", 15, 0.001, ["--table_ring", "--ti_ring"])
if "c4m" in which: run("c4-mid", 50000, 200, [0, 40, 80, 120, 160], "cgd", 2, 0.001, ["--width_phase2=32", "--prec_phase2=30", "--table_ring"], w2=32)
