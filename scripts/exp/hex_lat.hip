// column-split AES for latency-bound gate steps: 16 waves per record (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

#define DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))

// one AES column per lane: lane (bit, q) holds word q of the state of bit's block
__device__ __forceinline__ uint32_t aes_col(const LdsTab4 &lt, uint32_t s, const uint32_t *rkq) {
  s ^= rkq[0];
#pragma unroll
  for (int r = 1; r < 10; r++) {
    uint32_t v0 = lt.lkt(0, s, 0), v1 = lt.lkt(1, s, 1), v2 = lt.lkt(2, s, 2), v3 = lt.lkt(3, s, 3);
    s = v0 ^ rkq[r] ^ DPP(v1, 0x39) ^ DPP(v2, 0x4e) ^ DPP(v3, 0x93);
  }
  uint32_t u0 = lt.lkt(0, s, 0), u1 = lt.lkt(0, s, 1), u2 = lt.lkt(0, s, 2), u3 = lt.lkt(0, s, 3);
  uint32_t w0 = (u0 >> 8) & 0xffu, w1 = u1 & 0xff00u, w2 = u2 & 0xff0000u, w3 = (u3 << 16) & 0xff000000u;
  return w0 ^ rkq[10] ^ DPP(w1, 0x39) ^ DPP(w2, 0x4e) ^ DPP(w3, 0x93);
}

// VAR 0: 4 waves, one whole hash per wave (current 4-wave step, four-table AES)
// VAR 1: NW waves, wave v = hash (v >> 2), bit group (v & 3); one column per lane
template<int VAR, int NW>
__global__ void __launch_bounds__(NW * 64) lat_kernel(unsigned long long* out, int iters, uint32_t *check) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  __shared__ Lbl xch[2 * 256];
  __shared__ uint32_t stage[VAR ? NW * 256 : 1];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rkq[11];
  for (int r = 0; r < 11; r++) rkq[r] = c_rk[4 * r + (lane & 3)];
  Lbl x = {(uint32_t)lane * 2654435761u, 1u, 2u, 3u};
  Lbl R = {0x12345679u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const int NH = VAR ? NW / 4 : NW;
  for (int i = 0; i < iters; i++) {
    Lbl *xb = xch + (i & 1) * 256;
    if (VAR == 0) {
      Lbl in = (wave & 1) ? lxor(x, R) : x; uint64_t tw = 2 * (uint64_t)i + (wave >> 1); Lbl h;
      hash_n<1, LdsTab4>(lt, c_rk, &in, &tw, &h, c_rk24);
      xb[wave * 64 + lane] = h;
    } else {
      const int h = wave >> 2, g = wave & 3;
      Lbl in = (h & 1) ? lxor(x, R) : x; uint64_t tw = 2 * (uint64_t)i + (h >> 1);
      uint32_t k[4]; hash_prep(in, tw, k);
      uint32_t *st = stage + wave * 256;
      *reinterpret_cast<uint4 *>(st + lane * 4) = make_uint4(k[0], k[1], k[2], k[3]);
      uint32_t kq = st[(g * 16 + (lane >> 2)) * 4 + (lane & 3)];
      uint32_t o = aes_col(lt, kq, rkq) ^ kq;
      reinterpret_cast<uint32_t *>(xb)[(h * 64 + g * 16 + (lane >> 2)) * 4 + (lane & 3)] = o;
    }
    lds_barrier();
    Lbl acc = xb[lane];
    for (int q = 1; q < NH; q++) acc = lxor(acc, xb[q * 64 + lane]);
    x = acc;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (wave == 0) check[lane] = x.x ^ x.y ^ x.z ^ x.w;
}
template<int VAR, int NW> uint32_t run(const char* name) {
  unsigned long long* d; hipMalloc(&d, 64); uint32_t *c; hipMalloc(&c, 256);
  int iters = 2000;
  hipLaunchKernelGGL((lat_kernel<VAR, NW>), dim3(1), dim3(NW * 64), 0, 0, d, 10, c); hipDeviceSynchronize();
  hipLaunchKernelGGL((lat_kernel<VAR, NW>), dim3(1), dim3(NW * 64), 0, 0, d, iters, c); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  uint32_t hc[64]; hipMemcpy(hc, c, 256, hipMemcpyDeviceToHost);
  uint32_t x = 0; for (int i = 0; i < 64; i++) x = x * 31 + hc[i];
  printf("%-52s %7.0f cycles/step, %6.3f us/step (clock %.2f GHz) check %08x\n", name, (double)h[0]/iters, (double)h[1]/iters/100.0, (double)h[0]/((double)h[1]*10.0), x);
  return x;
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
  uint32_t a = run<0, 4>("4 waves: whole hash per wave (4 hashes)");
  uint32_t b = run<1, 16>("16 waves: one column per lane (4 hashes)");
  printf("same result: %s\n", a == b ? "yes" : "NO");
  uint32_t c2 = run<0, 2>("2 waves: whole hash per wave (2 hashes, evaluator)");
  uint32_t d2 = run<1, 8>("8 waves: one column per lane (2 hashes, evaluator)");
  printf("same result: %s\n", c2 == d2 ? "yes" : "NO");
  return 0;
}
