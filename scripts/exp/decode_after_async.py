"""how long lgc_party_decode_bits takes after a garbler drove its launches synchronously / asynchronously (diagnostic)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
for d, alg, it in ((100, "cgd", 15), (20, "cholesky", 0)):
    for mode in (0, 1, 2, 0, 2):
        sysm = lgc.make_system(d, 64, 56, alg, it, 0.001, 2, 1)
        G = lgc.Party(sysm, lgc.GARBLER, seed=bytes(range(16)))
        G.ring_create(64)
        n = G.num_launches
        t0 = time.perf_counter()
        if mode == 0:
            for k in range(n): G.garble_ring(k)
        else:
            if mode == 2: G.garble_ring_streams(1)
            for lo in range(0, n, 32):
                hi = min(n, lo + 32)
                for k in range(lo, hi): G.garble_ring_begin(k)
                for k in range(lo, hi): G.garble_ring_wait(k)
        t1 = time.perf_counter()
        G.decode_bits()
        t2 = time.perf_counter()
        G.decode_bits()
        t3 = time.perf_counter()
        print("d=%d %s mode %d: %d launches %.1f ms, decode_bits %.3f ms, again %.3f ms" % (d, alg, mode, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
        G.close()
