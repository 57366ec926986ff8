// diagnostic (not product code): cost of one cooperative gate level as a function of the number
// of waves in the workgroup, the waves that hash, and the hashes per wave (four-table AES image).
// Answers: how many hash-waves fit in one dependent level of a single record before the CU's AES
// throughput, not the latency of one hash, sets the level time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

template <int NW, int HA, int HPW, bool LOADS>
__global__ void __launch_bounds__(NW * 64) lat_kernel(unsigned long long *out, int iters, Lbl *gtab) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  __shared__ Lbl xch[2 * 1024];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Lbl x = {(uint32_t)lane, 1u, 2u, 3u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    Lbl *xb = xch + (i & 1) * 1024;
    Lbl tg = lzero(), te = lzero();
    if (LOADS) { tg = ld_lbl(gtab + (size_t)i * 128 + lane); te = ld_lbl(gtab + (size_t)i * 128 + 64 + lane); }
    if (wave < HA) {
      Lbl xs[HPW]; uint64_t tw[HPW]; Lbl h[HPW];
#pragma unroll
      for (int k = 0; k < HPW; k++) { xs[k] = x; xs[k].y ^= (uint32_t)(wave * 8 + k); tw[k] = (uint64_t)i * 64 + k; }
      hash_n<HPW, LdsTab4>(lt, c_rk, xs, tw, h, c_rk24);
#pragma unroll
      for (int k = 0; k < HPW; k++) xb[(wave * HPW + k) * 64 + lane] = h[k];
    }
    lds_barrier();
    Lbl acc = lxor(tg, te);
#pragma unroll
    for (int k = 0; k < 4; k++) acc = lxor(acc, xb[((k * 7) % (HA * HPW)) * 64 + lane]);
    x = acc;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = x.x; }
}
template <int NW, int HA, int HPW, bool LOADS> void run() {
  unsigned long long *d; hipMalloc(&d, 64);
  Lbl *g; hipMalloc(&g, (size_t)4096 * 128 * 16); hipMemset(g, 0x11, (size_t)4096 * 128 * 16);
  int iters = 2000;
  hipLaunchKernelGGL((lat_kernel<NW, HA, HPW, LOADS>), dim3(1), dim3(NW * 64), 0, 0, d, 10, g); hipDeviceSynchronize();
  hipLaunchKernelGGL((lat_kernel<NW, HA, HPW, LOADS>), dim3(1), dim3(NW * 64), 0, 0, d, iters, g); hipDeviceSynchronize();
  unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("waves %2d hashing %2d x %d = %2d hash-waves/level%s: %7.0f cycles/level %6.3f us (%.0f cycles per hash-wave)\n", NW, HA, HPW,
         HA * HPW, LOADS ? " +table loads" : "", (double)h[0] / iters, (double)h[1] / iters / 100.0, (double)h[0] / iters / (HA * HPW));
  hipFree(d); hipFree(g);
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
  run<1, 1, 1, false>();
  run<4, 1, 1, false>();
  run<4, 2, 1, false>();
  run<4, 4, 1, false>();
  run<4, 4, 2, false>();
  run<4, 4, 3, false>();
  run<8, 4, 1, false>();
  run<8, 8, 1, false>();
  run<8, 8, 2, false>();
  run<8, 8, 3, false>();
  run<12, 12, 1, false>();
  run<12, 12, 2, false>();
  run<16, 4, 1, false>();
  run<16, 8, 1, false>();
  run<16, 16, 1, false>();
  run<16, 12, 2, false>();
  run<16, 16, 2, false>();
  run<4, 2, 1, true>();
  run<4, 4, 1, true>();
  run<8, 4, 1, true>();
  run<8, 8, 1, true>();
  return 0;
}
