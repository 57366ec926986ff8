// diagnostic (not product code): master / helper gate step.  ONE wave walks the dependent chain (the circuit's
// glue); per level it stages the hash inputs in LDS, 4 * NH helper waves compute the hashes with one AES COLUMN per
// lane (a block spread over 4 lanes, 16 blocks per wave), and the master reads the results back: two barriers and
// two LDS round trips per level, but the AES itself is 4 lookups per lane per round.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;
#define DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))
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
// NHASH hashes per level; helper waves: HW (each takes ceil(4*NHASH/HW) (hash, group) items, interleaved)
template <int NHASH, int HW, int GLUE>
__global__ void __launch_bounds__((HW + 1) * 64) k(unsigned long long *out, int iters, uint32_t *check) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  __shared__ uint32_t stage[4 * 256], res[4 * 256];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t rkq[11];
  for (int r = 0; r < 11; r++) rkq[r] = c_rk[4 * r + (lane & 3)];
  Lbl x = {(uint32_t)lane * 2654435761u, 1u, 2u, 3u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  constexpr int ITEMS = 4 * NHASH, PER = (ITEMS + HW - 1) / HW;
  for (int i = 0; i < iters; i++) {
    if (wave == 0) {
#pragma unroll
      for (int h = 0; h < NHASH; h++) {
        Lbl in = x; in.y ^= (uint32_t)h; uint32_t kk[4]; hash_prep(in, 2 * (uint64_t)i + h, kk);
        *reinterpret_cast<uint4 *>(stage + h * 256 + lane * 4) = make_uint4(kk[0], kk[1], kk[2], kk[3]);
      }
    }
    lds_barrier();
    if (wave != 0) {
      uint32_t kq[PER], o[PER];
#pragma unroll
      for (int t = 0; t < PER; t++) {
        int item = (wave - 1) + t * HW;
        int h = item >> 2, g = item & 3;
        kq[t] = item < ITEMS ? stage[h * 256 + (g * 16 + (lane >> 2)) * 4 + (lane & 3)] : 0u;
      }
#pragma unroll
      for (int t = 0; t < PER; t++) o[t] = aes_col(lt, kq[t], rkq) ^ kq[t];
#pragma unroll
      for (int t = 0; t < PER; t++) {
        int item = (wave - 1) + t * HW;
        int h = item >> 2, g = item & 3;
        if (item < ITEMS) res[h * 256 + (g * 16 + (lane >> 2)) * 4 + (lane & 3)] = o[t];
      }
    }
    lds_barrier();
    if (wave == 0) {
      Lbl acc = lzero();
#pragma unroll
      for (int h = 0; h < NHASH; h++) { uint4 v = *reinterpret_cast<uint4 *>(res + h * 256 + lane * 4); Lbl l = {v.x, v.y, v.z, v.w}; acc = lxor(acc, l); }
      // stand-in for the circuit's glue between two gate steps: dependent lane shifts and selects
#pragma unroll
      for (int gq = 0; gq < GLUE; gq++) {
        Lbl y; int src = ((lane - 1 - gq) & 63) << 2;
        y.x = __builtin_amdgcn_ds_bpermute(src, acc.x); y.y = __builtin_amdgcn_ds_bpermute(src, acc.y);
        y.z = __builtin_amdgcn_ds_bpermute(src, acc.z); y.w = __builtin_amdgcn_ds_bpermute(src, acc.w);
        acc = lxor(acc, y);
      }
      x = acc;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (wave == 0) check[lane] = x.x ^ x.y ^ x.z ^ x.w;
}
template <int NHASH, int HW, int GLUE> void run() {
  unsigned long long *d; hipMalloc(&d, 64); uint32_t *c; hipMalloc(&c, 256);
  int iters = 2000;
  hipLaunchKernelGGL((k<NHASH, HW, GLUE>), dim3(1), dim3((HW + 1) * 64), 0, 0, d, 10, c); hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NHASH, HW, GLUE>), dim3(1), dim3((HW + 1) * 64), 0, 0, d, iters, c); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  uint32_t hc[64]; hipMemcpy(hc, c, 256, hipMemcpyDeviceToHost);
  uint32_t x = 0; for (int i = 0; i < 64; i++) x = x * 31 + hc[i];
  printf("master + %2d helper waves, %d hashes per level, glue %d shifts: %6.0f cycles/level %6.3f us  check %08x\n", HW, NHASH, GLUE,
         (double)h[0] / iters, (double)h[1] / iters / 100.0, x);
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  run<2, 8, 0>(); run<2, 8, 2>(); run<2, 4, 2>();
  run<4, 8, 2>(); run<4, 15, 2>(); run<4, 12, 2>();
  run<2, 15, 2>();
  return 0;
}
