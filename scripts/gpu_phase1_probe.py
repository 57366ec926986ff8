"""phase-1 kernel throughput probe (config 3/4 shapes)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
rng = np.random.default_rng(0)
for (n, d, c0, c1) in ((10000, 100, 0, 50), (50000, 500, 0, 100), (50000, 500, 400, 500), (1000000, 100, 0, 50)):
    Xq = rng.integers(-2**27, 2**27, size=(n, d), dtype=np.int64); yq = rng.integers(-2**27, 2**27, size=n, dtype=np.int64)
    t0 = time.perf_counter(); ph = lgc.Phase1(Xq, yq, 64, 56); t1 = time.perf_counter()
    for rep in range(2):
        t2 = time.perf_counter(); A, b = ph.local(c0, c1, with_y=True); t3 = time.perf_counter()
    own = c1 - c0
    macs = n * (own + 1) * (own + 2) / 2
    # check a few entries on the CPU
    i, j = own - 1, own // 2
    exp = int((Xq[:, c0 + i].astype(object) * Xq[:, c0 + j].astype(object)).sum()) & (2**64 - 1)
    ok = int(A[i * (i + 1) // 2 + j]) == exp
    print("n=%d d=%d cols [%d,%d): upload %.3fs local block %.4fs -> %.3e u64 MAC/s (incl. host round trip) ok=%s" % (n, d, c0, c1, t1 - t0, t3 - t2, macs / (t3 - t2), ok), flush=True)
    ph.close()
for (npairs, n) in ((64, 10000), (16, 50000)):
    for rep in range(2):
        t0 = time.perf_counter(); x, y, r, xyr = lgc.ti_generate(bytes(range(16)), 0, npairs, n, 64); t1 = time.perf_counter()
    print("TI generate npairs=%d n=%d: %.4fs -> %.3e words/s" % (npairs, n, t1 - t0, npairs * (2 * n + 1) / (t1 - t0)), flush=True)
