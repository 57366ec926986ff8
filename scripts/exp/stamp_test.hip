// diagnostic: cycle breakdown of single gate steps in the 4-wave kernel on an OP_MAX-like chain of adds
#define GC_STAMP 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  const size_t NT = 65536 * 4;
  unsigned long long* d_st; hipMalloc(&d_st, NT * 8);
  Lbl* words; hipMalloc(&words, 64*1024); hipMemset(words, 0x5a, 64*1024);
  Lbl* tab; hipMalloc(&tab, (size_t)4096*2048); hipMemset(tab, 0x33, (size_t)4096*2048);
  uint64_t* dec; hipMalloc(&dec, 64);
  Rec r; r.op = OP_DIV; r.cnt = 1; r.dst = 20; r.a = 1; r.b = 2; r.c = 0; r.sa = 1; r.sb = 1; r.step0 = 0;
  Rec* d_r; hipMalloc(&d_r, sizeof(Rec)); hipMemcpy(d_r, &r, sizeof(Rec), hipMemcpyHostToDevice);
  Lbl R = {1,2,3,4};
  static unsigned long long h[NT];
  for (int g = 0; g < 2; g++) {
    hipMemset(d_st, 0, NT * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &d_st, sizeof(d_st));
    if (g == 0) hipLaunchKernelGGL((gc_exec_kernel<true, true>), dim3(1), dim3(256), 0, 0, d_r, 1u, words, tab, dec, 0ull, R, 64, 56);
    else hipLaunchKernelGGL((gc_exec_kernel<false, true>), dim3(1), dim3(256), 0, 0, d_r, 1u, words, tab, dec, 0ull, R, 64, 56);
    hipDeviceSynchronize();
    hipMemcpy(h, d_st, NT * 8, hipMemcpyDeviceToHost);
    // steps are numbered by their first gate step; dual steps leave the odd slot empty
    double hash1 = 0, bar1 = 0, tail1 = 0, glue1 = 0, n1 = 0, hash2 = 0, bar2 = 0, tail2 = 0, glue2 = 0, n2 = 0;
    unsigned long long prev_exit = 0; int prev_kind = 0;
    for (size_t s = 0; s < 1596; s++) {
      unsigned long long* t = h + 4 * s;
      if (!t[0]) continue;
      bool dual = (s + 1 < 1596) && (h[4 * (s + 1)] == 0);
      double hs = (double)(t[1] - t[0]), br = (double)(t[2] - t[1]), tl = (double)(t[3] - t[2]);
      double gl = prev_exit ? (double)(t[0] - prev_exit) : 0;
      if (dual) { hash2 += hs; bar2 += br; tail2 += tl; glue2 += gl; n2++; } else { hash1 += hs; bar1 += br; tail1 += tl; glue1 += gl; n1++; }
      prev_exit = t[3];
    }
    unsigned long long first = 0, last = 0; for (size_t s = 0; s < 1596; s++) if (h[4*s]) { if (!first) first = h[4*s]; last = h[4*s+3]; }
    printf("%s DIV: total %.0f cycles; single steps %.0f: hash %.0f barrier %.0f tail %.0f glue-before %.0f | dual steps %.0f: hash %.0f barrier %.0f tail %.0f glue-before %.0f\n",
           g == 0 ? "garbler" : "evaluator", (double)(last - first), n1, hash1/n1, bar1/n1, tail1/n1, glue1/n1, n2, hash2/n2, bar2/n2, tail2/n2, glue2/n2);
  }
  return 0;
}
