// diagnostic: cycle breakdown of single gate steps in the 4-wave kernel on an OP_MAX-like chain of adds
#define GC_STAMP 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  unsigned long long* d_st; hipMalloc(&d_st, 64); 
  Lbl* words; hipMalloc(&words, 64*1024); hipMemset(words, 0x5a, 64*1024);
  Lbl* tab; hipMalloc(&tab, (size_t)4096*2048); hipMemset(tab, 0x33, (size_t)4096*2048);
  uint64_t* dec; hipMalloc(&dec, 64);
  Rec r; r.op = OP_MAX; r.cnt = 8; r.dst = 20; r.a = 1; r.b = 0; r.c = 0; r.sa = 1; r.sb = 1; r.step0 = 0;
  Rec* d_r; hipMalloc(&d_r, sizeof(Rec)); hipMemcpy(d_r, &r, sizeof(Rec), hipMemcpyHostToDevice);
  Lbl R = {1,2,3,4};
  for (int g = 0; g < 2; g++) {
    hipMemset(d_st, 0, 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &d_st, sizeof(d_st));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    if (g == 0) hipLaunchKernelGGL((gc_exec_kernel<true, true>), dim3(1), dim3(256), 0, 0, d_r, 1u, words, tab, dec, 0ull, R, 64, 56);
    else hipLaunchKernelGGL((gc_exec_kernel<false, true>), dim3(1), dim3(256), 0, 0, d_r, 1u, words, tab, dec, 0ull, R, 64, 56);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8]; hipMemcpy(h, d_st, 64, hipMemcpyDeviceToHost);
    double n = (double)h[4];
    printf("%s: kernel %.3f ms, single steps %.0f: hash %.0f  barrier %.0f  glue-before %.0f  tail %.0f cycles/step\n", g == 0 ? "garbler" : "evaluator", ms, n, h[0]/n, h[1]/n, h[2]/n, h[5]/n);
  }
  return 0;
}
