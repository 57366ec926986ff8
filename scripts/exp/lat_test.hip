// latency decomposition of one cooperative gate step (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

template<int VAR>
__global__ void __launch_bounds__(256) lat_kernel(unsigned long long* out, int iters, Lbl* gtab) {
  __shared__ uint32_t lds_te0[kLdsTabWords];
  __shared__ Lbl xch[2*512];
  lds_tab_fill(lds_te0);
  LdsTab lt = lds_tab_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Lbl x = {threadIdx.x, 1u, 2u, 3u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (VAR == 0) {            // dependent chain of single hashes, one wave
      uint64_t tw = i; Lbl h; hash_n<1, LdsTab>(lt, c_rk, &x, &tw, &h); x = h;
    } else if (VAR == 1) {     // + LDS exchange + barrier (4 waves)
      uint64_t tw = i; Lbl h; hash_n<1, LdsTab>(lt, c_rk, &x, &tw, &h);
      xch[(i&1)*512 + wave*64 + lane] = h; __syncthreads();
      Lbl h0 = xch[(i&1)*512 + lane], h1 = xch[(i&1)*512 + 64 + lane], h2 = xch[(i&1)*512 + 128+lane], h3 = xch[(i&1)*512+192+lane];
      x = lxor(lxor(h0,h1), lxor(h2,h3));
    } else if (VAR == 2) {     // + global table store (wave 0) and 2 bpermute shifts
      uint64_t tw = i; Lbl h; hash_n<1, LdsTab>(lt, c_rk, &x, &tw, &h);
      xch[(i&1)*512 + wave*64 + lane] = h; __syncthreads();
      Lbl h0 = xch[(i&1)*512 + lane], h1 = xch[(i&1)*512 + 64 + lane], h2 = xch[(i&1)*512 + 128+lane], h3 = xch[(i&1)*512+192+lane];
      x = lxor(lxor(h0,h1), lxor(h2,h3));
      if (wave == 0) { st_lbl(gtab + (size_t)i*128 + lane, x); st_lbl(gtab + (size_t)i*128 + 64 + lane, h0); }
      Lbl y; y.x = __builtin_amdgcn_ds_bpermute(((lane-1)&63)<<2, x.x); y.y = __builtin_amdgcn_ds_bpermute(((lane-1)&63)<<2, x.y);
      y.z = __builtin_amdgcn_ds_bpermute(((lane-1)&63)<<2, x.z); y.w = __builtin_amdgcn_ds_bpermute(((lane-1)&63)<<2, x.w);
      x = lxor(x, y);
    } else if (VAR == 3) {     // two interleaved hashes per wave (dual step, garbler)
      Lbl xs[2] = {x, lxor(x, x)}; xs[1].x = i; uint64_t tw[2] = {(uint64_t)i, (uint64_t)i+1}; Lbl h[2];
      hash_n<2, LdsTab>(lt, c_rk, xs, tw, h);
      xch[(i&1)*512 + wave*64 + lane] = h[0]; xch[(i&1)*512 + 256 + wave*64 + lane] = h[1]; __syncthreads();
      Lbl h0 = xch[(i&1)*512 + lane], h1 = xch[(i&1)*512 + 64 + lane], h2 = xch[(i&1)*512 + 128+lane], h3 = xch[(i&1)*512+192+lane];
      x = lxor(lxor(h0,h1), lxor(h2,h3));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = x.x; }
}
template<int VAR> void run(const char* name, int threads) {
  unsigned long long* d; hipMalloc(&d, 64); Lbl* g; hipMalloc(&g, (size_t)4096*128*16);
  int iters = 2000;
  hipLaunchKernelGGL((lat_kernel<VAR>), dim3(1), dim3(threads), 0, 0, d, 10, g); hipDeviceSynchronize();
  hipLaunchKernelGGL((lat_kernel<VAR>), dim3(1), dim3(threads), 0, 0, d, iters, g); hipDeviceSynchronize();
  unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%-46s %7.0f shader cycles/step, %6.3f us/step (clock %.2f GHz)\n", name, (double)h[0]/iters, (double)h[1]/iters/100.0, (double)h[0]/((double)h[1]*10.0));
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  run<0>("1 wave: dependent hash chain", 64);
  run<0>("4 waves: dependent hash chain each", 256);
  run<1>("4 waves: hash + LDS exchange + barrier", 256);
  run<2>("4 waves: + table store + bpermute shift", 256);
  run<3>("4 waves: 2 hashes/wave + exchange + barrier", 256);
  return 0;
}
