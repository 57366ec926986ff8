"""where the creation of a 64-lambda sweep solver goes (LINREG_TRACE=1 for the library's marks) -- diagnostic"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import linreg_gc as lgc
d, nl = 100, 64
sysm = lgc.make_system(d, 64, 56, "cgd", 15, 0.0, 2, 1)
lam = np.linspace(0.001, 1.0, nl)
lgc.device_count()
w = lgc.Solver(lgc.make_system(5, 64, 56, "cgd", 1, 0.0, 2, 1)); w.close()      # context, kernels
for rep in range(3):
    t0 = time.perf_counter()
    P = lgc.Program(sysm, lambdas=lam)
    t1 = time.perf_counter()
    P.close()
    s = lgc.Solver(sysm, lambdas=lam)
    t2 = time.perf_counter()
    print("program alone %.3f s, solver (program + device state) %.3f s" % (t1 - t0, t2 - t1), flush=True)
    s.close()
