"""diagnostic: solver creation beyond device memory must come back as an error, not a crash"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import torch  # noqa: F401
import linreg_gc as lgc
for d in (2000, 4000, 8000):
    try:
        s = lgc.Solver(lgc.make_system(d, 64, 56, "cgd", 2, 0.0, 2, 0, 0, 0))
        print("d", d, "created", flush=True)
        s.close()
    except Exception as e:
        print("d", d, "->", type(e).__name__, str(e)[:200], flush=True)
print("alive")
