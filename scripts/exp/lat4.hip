// diagnostic (not product code): one cooperative gate level with the AES state of a block spread over the four
// lanes of a quad (lane 4j + c holds column c of block j; ShiftRows = three quad_perm DPP moves), so that a wave
// issues 4 lookups per round instead of 16 and a 64-gate hash takes four waves.  Question: does the level get
// shorter than the 2914 cycles of lat2's "4 waves x 1 whole-block hash" when 16 waves share the work?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

template <int CTRL> __device__ __forceinline__ uint32_t qp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// lane 4j + c: column c of sigma(x_j) ^ tweak_j hashed; returns column c of H
__device__ __forceinline__ uint32_t hash_split(const LdsTab4 &lt, const uint32_t rkl[11], uint32_t x, uint32_t twc, int c) {
  uint32_t t = qp<0x4E>(x);
  uint32_t k = (c < 2) ? (t ^ twc) : (x ^ t);
  uint32_t s = k ^ rkl[0];
#pragma unroll
  for (int rnd = 1; rnd < 10; rnd++) {
    uint32_t u0 = lt.lkt(0, s, 0), u1 = lt.lkt(1, s, 1), u2 = lt.lkt(2, s, 2), u3 = lt.lkt(3, s, 3);
    s = (u0 ^ rkl[rnd]) ^ qp<0x39>(u1) ^ qp<0x4E>(u2) ^ qp<0x93>(u3);
  }
  uint32_t s1 = qp<0x39>(s), s2 = qp<0x4E>(s), s3 = qp<0x93>(s);
  uint32_t v0 = lt.lk(s, 0), v1 = lt.lk(s1, 1), v2 = lt.lk(s2, 2), v3 = lt.lk(s3, 3);
  s = xor3(last_lo(v1, v0), last_hi(v3, v2), rkl[10]);
  return s ^ k;
}

// correctness: 64 labels hashed both ways
__global__ void __launch_bounds__(256) check_kernel(uint32_t *bad) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 3, j = lane >> 2;
  uint32_t rkl[11];
  for (int r = 0; r < 11; r++) rkl[r] = c_rk[4 * r + c];
  // label g = 16 * wave + j
  const int g = 16 * wave + j;
  Lbl x = {0x01234567u * (g + 1), 0x89abcdefu ^ (g * 77u), 0xdeadbeefu + g, 0x13579bdfu * (g + 3)};
  uint64_t tw = 0x1122334455667788ull + (uint64_t)g * 0x100000001ull;
  Lbl h;
  hash_n<1, LdsTab4>(lt, c_rk, &x, &tw, &h, c_rk24);
  uint32_t xc = c == 0 ? x.x : c == 1 ? x.y : c == 2 ? x.z : x.w;
  uint32_t twc = c == 0 ? (uint32_t)tw : c == 1 ? (uint32_t)(tw >> 32) : 0u;
  uint32_t hs = hash_split(lt, rkl, xc, twc, c);
  uint32_t hc = c == 0 ? h.x : c == 1 ? h.y : c == 2 ? h.z : h.w;
  if (hs != hc) atomicAdd(bad, 1u);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) lat_kernel(unsigned long long *out, int iters) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  __shared__ uint32_t xch[2 * 1024];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 3;
  const int q = wave >> 2, r = wave & 3;       // hash id, gate block
  uint32_t rkl[11];
  for (int k = 0; k < 11; k++) rkl[k] = c_rk[4 * k + c];
  uint32_t x = (uint32_t)lane * 2654435761u + wave;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    uint32_t *xb = xch + (i & 1) * 1024;
    uint32_t twc = c == 0 ? (uint32_t)(i * 64 + q) : 0u;
    uint32_t h = hash_split(lt, rkl, x ^ (uint32_t)q, twc, c);
    xb[q * 256 + r * 64 + lane] = h;
    lds_barrier();
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) acc ^= xb[((k * 3) % (NW / 4)) * 256 + r * 64 + lane];
    x = acc;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = x; }
}
template <int NW> void run() {
  unsigned long long *d; hipMalloc(&d, 64);
  int iters = 2000;
  hipLaunchKernelGGL((lat_kernel<NW>), dim3(1), dim3(NW * 64), 0, 0, d, 10); hipDeviceSynchronize();
  hipLaunchKernelGGL((lat_kernel<NW>), dim3(1), dim3(NW * 64), 0, 0, d, iters); hipDeviceSynchronize();
  unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("split hash: %2d waves = %d hashes of 64 gates per level: %7.0f cycles/level %6.3f us\n", NW, NW / 4, (double)h[0] / iters,
         (double)h[1] / iters / 100.0);
  hipFree(d);
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
  hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
  uint32_t *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(256), 0, 0, bad); hipDeviceSynchronize();
  uint32_t hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("split hash vs whole-block hash: %u mismatching columns of 256\n", hb);
  run<4>(); run<8>(); run<16>();
  return 0;
}
