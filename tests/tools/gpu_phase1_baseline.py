"""Phase-1 local aggregation (inner_product_local + the floating-point diagonal, src/phase1.c:14-20,
562-571): GPU kernels against the single-threaded CPU oracle on the same quantised data.
    python tests/tools/gpu_phase1_baseline.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linreg_gc as lgc, orc
oracle = orc.load()
rng = np.random.default_rng(0)
for (n, d, cpu) in ((10000, 100, True), (50000, 100, True), (50000, 500, True), (1000000, 100, False)):
    X = rng.standard_normal((n, d)); X /= np.abs(X).max(axis=0); y = rng.standard_normal(n)
    Xq = np.asarray(oracle.quantize(X, 56, n, 64)).reshape(n, d); yq = np.asarray(oracle.quantize(y, 56, n, 64))
    ph = lgc.Phase1(Xq, yq, 64, 56)
    ph.local(0, d, with_y=True)
    t0 = time.perf_counter(); A, b = ph.local(0, d, with_y=True); tg = time.perf_counter() - t0
    macs = n * (d + 1) * (d + 2) / 2
    rec = dict(n=n, d=d, u64_macs=macs, gpu_call_s=round(tg, 4), gpu_macs_per_s=macs / tg)
    if cpu:
        t0 = time.perf_counter(); Ao, bo = oracle.aggregate(Xq, yq, n, d, 56, 64); tc = time.perf_counter() - t0
        rec.update(cpu_oracle_s=round(tc, 3), cpu_macs_per_s=macs / tc, speedup=tc / tg,
                   exact=bool(np.array_equal(np.asarray(A, dtype=np.uint64), np.asarray(Ao, dtype=np.uint64)) and
                              np.array_equal(np.asarray(b, dtype=np.uint64), np.asarray(bo, dtype=np.uint64))))
    print(json.dumps(rec), flush=True)
    ph.close()
