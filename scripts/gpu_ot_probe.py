"""OT-extension throughput through the C ABI (C3 shape: n = 10^4, 64-bit: 6.4e5 OTs per inner product), three ways:
  pageable  numpy buffers (what a caller that knows nothing about the GPU passes)
  pinned    page-locked buffers from lgc_host_alloc: async copies on the session stream at PCIe rate
  device    lgc_ot_*_set_device_io(1): operands, u and y stay in HBM (same-node hand-off)
Per OT the ABI moves 16 B of u and 8 B of y in each direction over PCIe in the host modes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
import numpy as np
import torch
import linreg_gc as lgc
rng = np.random.default_rng(0)
seeds0 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8); seeds1 = rng.integers(0, 256, size=(128, 16), dtype=np.uint8)
delta = rng.integers(0, 256, size=16, dtype=np.uint8)
dbits = np.unpackbits(delta, bitorder="little")
n, w = 10000, 64
M64 = 2**64 - 1


def check(ss, sr, a, b):
    return all(((int(x) + int(z)) & M64) == (sum(int(p) * int(q) for p, q in zip(ra, rb)) & M64) for x, z, ra, rb in zip(ss[:2], sr[:2], a[:2], b[:2]))


for npairs in (4, 64):
    m = npairs * n * w
    a = rng.integers(0, 2**63, size=(npairs, n), dtype=np.uint64); b = rng.integers(0, 2**63, size=(npairs, n), dtype=np.uint64)
    # ---- pageable
    S = lgc.OtSender(delta.tobytes(), np.where(dbits[:, None] == 1, seeds1, seeds0)); R = lgc.OtReceiver(seeds0, seeds1)
    for rep in range(3):
        t0 = time.perf_counter(); u = R.gilboa_start(a, w); y, ss = S.gilboa(b, w, u); sr = R.gilboa_finish(y); t3 = time.perf_counter()
    print("npairs=%d m=%.2e pageable host buffers: %.3e OT/s ok=%s" % (npairs, m, m / (t3 - t0), check(ss, sr, a, b)), flush=True)
    S.close(); R.close()
    # ---- pinned
    S = lgc.OtSender(delta.tobytes(), np.where(dbits[:, None] == 1, seeds1, seeds0)); R = lgc.OtReceiver(seeds0, seeds1)
    ub = lgc.lib().lgc_ot_u_bytes(m)
    pa, pb, pu, py = lgc.host_alloc(a.nbytes), lgc.host_alloc(b.nbytes), lgc.host_alloc(ub), lgc.host_alloc(m * 8)
    pss, psr = lgc.host_alloc(npairs * 8), lgc.host_alloc(npairs * 8)
    pa[:] = a.view(np.uint8).ravel(); pb[:] = b.view(np.uint8).ravel()
    for rep in range(3):
        t0 = time.perf_counter()
        R.gilboa_start_ptr(pa.ctypes.data, npairs, n, w, pu.ctypes.data)
        S.gilboa_ptr(pb.ctypes.data, npairs, n, w, pu.ctypes.data, py.ctypes.data, pss.ctypes.data)
        R.gilboa_finish_ptr(py.ctypes.data, psr.ctypes.data)
        t3 = time.perf_counter()
    print("npairs=%d m=%.2e pinned host buffers:   %.3e OT/s ok=%s" % (npairs, m, m / (t3 - t0), check(pss.view(np.uint64), psr.view(np.uint64), a, b)), flush=True)
    for x in (pa, pb, pu, py, pss, psr): lgc.host_free(x)
    S.close(); R.close()
    # ---- device-resident
    S = lgc.OtSender(delta.tobytes(), np.where(dbits[:, None] == 1, seeds1, seeds0)); R = lgc.OtReceiver(seeds0, seeds1)
    S.set_device_io(True); R.set_device_io(True)
    da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
    du = torch.empty(ub, dtype=torch.uint8, device="cuda"); dy = torch.empty(m, dtype=torch.int64, device="cuda")
    dss = torch.zeros(npairs, dtype=torch.int64, device="cuda"); dsr = torch.zeros(npairs, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        R.gilboa_start_ptr(da.data_ptr(), npairs, n, w, du.data_ptr())
        S.gilboa_ptr(db.data_ptr(), npairs, n, w, du.data_ptr(), dy.data_ptr(), dss.data_ptr())
        R.gilboa_finish_ptr(dy.data_ptr(), dsr.data_ptr())
        t3 = time.perf_counter()
    print("npairs=%d m=%.2e device-resident I/O:   %.3e OT/s ok=%s" % (npairs, m, m / (t3 - t0),
          check(dss.cpu().numpy().view(np.uint64), dsr.cpu().numpy().view(np.uint64), a, b)), flush=True)
    S.close(); R.close()
