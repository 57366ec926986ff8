import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
L = lgc.lib(); L.lgc_debug_spin.argtypes = [C.c_int, C.c_int, C.c_double]
d = 100; T = d * (d + 1) // 2
rng = np.random.default_rng(0)
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
sysm = lgc.make_system(d, 64, 56, "cgd", 15, 0.0, 2, 0, 0, 0)
s = lgc.Solver(sysm); s.set_shares(shares)
for blocks in (0, 16, 64, 256, 1024):
    for rep in range(2):
        if blocks: L.lgc_debug_spin(0, blocks, 700.0)
        t0 = time.perf_counter(); s.run(); t1 = time.perf_counter()
        time.sleep(0.8)
    print("spin blocks=%d: solve %.3fs (device %.3fs)" % (blocks, t1 - t0, s.stats()["seconds_total"]), flush=True)
