"""quick GPU probe: AES micro-benchmark + solver timings (not a test)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc

if os.environ.get("LGC_KARATSUBA") == "0":      # A/B: plain 64 x 64 array in the matrix-vector products
    lgc.set_karatsuba(False)
print("devices", lgc.device_count(), flush=True)
for waves, bpl in ((4096, 64), (16384, 256), (65536, 256)):
    r, c = lgc.aes_bench(waves, bpl)
    print("aes_bench waves=%d bpl=%d: %.3e blocks/s chk=%08x" % (waves, bpl, r, c), flush=True)

def run(d, alg, iters, w=64, p=56, profile=False):
    rng = np.random.default_rng(0)
    T = d * (d + 1) // 2
    shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
    sysm = lgc.make_system(d, w, p, alg, iters, 0.0, 2, 0, 0, 0)
    t0 = time.time()
    s = lgc.Solver(sysm)
    t1 = time.time()
    s.set_shares(shares)
    s.run(profile=profile)
    t2 = time.time()
    st = s.stats()
    print("d=%d %s-%d w=%d: create %.2fs run %.3fs dev %.3fs gates %.3e -> %.3e AND/s; mac G %.3fs E %.3fs mac_gates %.3e launches %d" % (
        d, alg, iters, w, t1 - t0, t2 - t1, st["seconds_total"], st["and_gates"], st["and_gates"] / st["seconds_total"],
        st["seconds_mac_garble"], st["seconds_mac_eval"], st["mac_gates"], st["launches"]), flush=True)
    if profile:
        print("   profiled: garble %.3fs eval %.3fs" % (st["seconds_garble"], st["seconds_eval"]), flush=True)
    s.close()

args = sys.argv[1:] or ["small"]
if "small" in args:
    run(20, "cgd", 2)
    run(100, "cgd", 2)
    run(100, "cgd", 2, profile=True)
    run(20, "cholesky", 0)
if "big" in args:
    run(500, "cgd", 1)
    run(500, "cgd", 1, profile=True)
if "big3" in args:     # the headline's shape, shorter: overlapped and serialised
    run(500, "cgd", 3)
    run(500, "cgd", 3)
    run(500, "cgd", 3, profile=True)
if "w32" in args:
    run(100, "cgd", 2, w=32, p=30)
if "mid" in args:      # overlap of the garbler and evaluator chains on latency-bound circuits
    for prof in (False, True, False):
        run(100, "cgd", 15, profile=prof)
    for prof in (False, True):
        run(20, "cholesky", 0, profile=prof)
if "chol" in args:     # latency-bound launches only (4-wave kernels)
    for _ in range(3):
        run(20, "cholesky", 0)
if "trace" in args:
    run(100, "cgd", 2)
if "trace500" in args:  # timeline of the two chains around the MAC launches (scripts/dbg/timeline.py reads the kernel trace)
    run(500, "cgd", 3)
    run(500, "cgd", 3)
if "chol250" in args:
    for prof in (False, True, False):
        run(250, "cholesky", 0, profile=prof)
if "trace100" in args:
    run(100, "cgd", 15)
    run(100, "cgd", 15)
if "w32big" in args:
    run(500, "cgd", 20, w=32, p=30)
    run(500, "cgd", 20, w=32, p=30)
