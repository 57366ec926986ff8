"""OT-extension throughput probe (C3 shape: n = 10^4, 64-bit: 6.4e5 OTs per inner product)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
rng = np.random.default_rng(0)
seeds0 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8); seeds1 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
delta = rng.integers(0, 256, size=16, dtype=np.uint8)
dbits = np.unpackbits(delta, bitorder="little")
S = lgc.OtSender(delta.tobytes(), np.where(dbits[:, None] == 1, seeds1, seeds0)); R = lgc.OtReceiver(seeds0, seeds1)
n, w = 10000, 64
for npairs in (4, 32, 64):
    a = rng.integers(0, 2**63, size=(npairs, n), dtype=np.uint64); b = rng.integers(0, 2**63, size=(npairs, n), dtype=np.uint64)
    for rep in range(2):
        t0 = time.perf_counter(); u = R.gilboa_start(a, w); t1 = time.perf_counter()
        y, ss = S.gilboa(b, w, u); t2 = time.perf_counter()
        sr = R.gilboa_finish(y); t3 = time.perf_counter()
    m = npairs * n * w
    ok = all(((int(x) + int(z)) & (2**64 - 1)) == (sum(int(p) * int(q) for p, q in zip(ra, rb)) & (2**64 - 1)) for x, z, ra, rb in zip(ss[:2], sr[:2], a[:2], b[:2]))
    print("npairs=%d m=%.2e OTs: recv_start %.3fs send %.3fs recv_finish %.3fs -> %.3e OT/s end-to-end (host buffers) ok=%s" % (
        npairs, m, t1 - t0, t2 - t1, t3 - t2, m / (t3 - t0), ok), flush=True)
