"""Time in the Karatsuba MAC kernels by wave index (library built with -DGC_MAC_WAVE_TRACE=1: scripts/exp/build_variant.sh
wavetrace -DGC_MAC_WAVE_TRACE=1; LGC_LIB=scripts/exp/libs/lib_wavetrace.so).  One d = 500 CGD iteration, serialised."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc

L = lgc.lib()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(0)
T = d * (d + 1) // 2
shares = rng.integers(0, 2**62, size=(2, T + d), dtype=np.uint64)
s = lgc.Solver(lgc.make_system(d, 64, 56, "cgd", 1, 0.0, 2, 0, 0, 0))
s.set_shares(shares)
s.run(profile=True)                       # warm
for f in (L.lgc_dbg_mac_wave_ticks_g, L.lgc_dbg_mac_wave_ticks_e):
    f(None, None, 1)
s.run(profile=True)
st = s.stats()
print("d=%d CGD-1 serialised: MAC garble %.4f s, evaluate %.4f s; LGC_MAC_CHUNK=%s" % (d, st["seconds_mac_garble"], st["seconds_mac_eval"], os.environ.get("LGC_MAC_CHUNK", "default")))
for name, f in (("garbler", L.lgc_dbg_mac_wave_ticks_g), ("evaluator", L.lgc_dbg_mac_wave_ticks_e)):
    t = (C.c_uint64 * 32)(); c = (C.c_uint64 * 32)()
    f(t, c, 0)
    off = 0 if name == "garbler" else 16
    vals = ["%.3f" % (t[off + w] / 1e5 / c[off + w]) if c[off + w] else "-" for w in range(16)]
    print("  %-9s mean ms in the kernel by wave index (workgroups: %d): %s" % (name, c[off], " ".join(vals)))
s.close()
