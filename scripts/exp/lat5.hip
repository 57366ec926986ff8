// diagnostic (not product code): gate level with the AES state of a block spread over TWO lanes (lane 2j + h holds
// columns 2h, 2h + 1 of block j; the partner's two columns arrive by one quad_perm swap each): 8 lookups and 14 vector
// instructions per round and lane -- between the whole-block form (16 lookups, 24 instructions: lat2.hip) and the
// four-lane form (4 lookups, 9 instructions: lat4.hip).  A 64-gate hash takes two waves; a 4-hash level eight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

template <int CTRL> __device__ __forceinline__ uint32_t qp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// lane 2j + h: (x0, x1) = columns 2h, 2h + 1 of label j; tw = tweak (both lanes hold it).  Returns the two columns of H.
__device__ __forceinline__ void hash_pair(const LdsTab4 &lt, const uint32_t *rk, uint32_t x0, uint32_t x1, uint64_t tw, int h,
                                          uint32_t &o0, uint32_t &o1) {
  // sigma(x) ^ t: k0 = x.z ^ tlo, k1 = x.w ^ thi, k2 = x.z ^ x.x, k3 = x.w ^ x.y
  const uint32_t p0 = qp<0xB1>(x0), p1 = qp<0xB1>(x1);      // partner's columns (quad_perm [1,0,3,2])
  uint32_t k0, k1;
  if (h == 0) { k0 = p0 ^ (uint32_t)tw; k1 = p1 ^ (uint32_t)(tw >> 32); }       // lane 0 holds x.x, x.y; partner z, w
  else        { k0 = x0 ^ p0;           k1 = x1 ^ p1; }                          // lane 1 holds x.z, x.w; partner x, y
  uint32_t s0 = k0 ^ rk[2 * h], s1 = k1 ^ rk[2 * h + 1];
#pragma unroll
  for (int rnd = 1; rnd < 10; rnd++) {
    const uint32_t q0 = qp<0xB1>(s0), q1 = qp<0xB1>(s1);    // partner columns: (2h+2, 2h+3) mod 4
    // new column c = T0[col c .b0] ^ T1[col c+1 .b1] ^ T2[col c+2 .b2] ^ T3[col c+3 .b3]
    // own columns a = 2h (s0), a+1 (s1); partner a+2 (q0), a+3 (q1)
    const uint32_t n0 = xor3(xor3(lt.lkt(0, s0, 0), lt.lkt(1, s1, 1), rk[4 * rnd + 2 * h]), lt.lkt(2, q0, 2), lt.lkt(3, q1, 3));
    const uint32_t n1 = xor3(xor3(lt.lkt(0, s1, 0), lt.lkt(1, q0, 1), rk[4 * rnd + 2 * h + 1]), lt.lkt(2, q1, 2), lt.lkt(3, s0, 3));
    s0 = n0; s1 = n1;
  }
  const uint32_t q0 = qp<0xB1>(s0), q1 = qp<0xB1>(s1);
  const uint32_t a0 = lt.lk(s0, 0), a1 = lt.lk(s1, 1), a2 = lt.lk(q0, 2), a3 = lt.lk(q1, 3);
  const uint32_t b0 = lt.lk(s1, 0), b1 = lt.lk(q0, 1), b2 = lt.lk(q1, 2), b3 = lt.lk(s0, 3);
  o0 = xor3(last_lo(a1, a0), last_hi(a3, a2), rk[40 + 2 * h]) ^ k0;
  o1 = xor3(last_lo(b1, b0), last_hi(b3, b2), rk[40 + 2 * h + 1]) ^ k1;
}

__global__ void __launch_bounds__(128) check_kernel(uint32_t *bad) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane & 1, j = lane >> 1;
  uint32_t rk[44];
  for (int i = 0; i < 44; i++) rk[i] = c_rk[i];
  const int g = 32 * wave + j;
  Lbl x = {0x01234567u * (g + 1), 0x89abcdefu ^ (g * 77u), 0xdeadbeefu + g, 0x13579bdfu * (g + 3)};
  uint64_t tw = 0x1122334455667788ull + (uint64_t)g * 0x100000001ull;
  Lbl hh;
  hash_n<1, LdsTab4>(lt, c_rk, &x, &tw, &hh, c_rk24);
  uint32_t o0, o1;
  hash_pair(lt, rk, h ? x.z : x.x, h ? x.w : x.y, tw, h, o0, o1);
  if (o0 != (h ? hh.z : hh.x)) atomicAdd(bad, 1u);
  if (o1 != (h ? hh.w : hh.y)) atomicAdd(bad, 1u);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) lat_kernel(unsigned long long *out, int iters) {
  __shared__ uint32_t lds_te0[2 * kLdsTabWords];
  __shared__ uint32_t xch[2 * 2048];
  lds_tab4_fill(lds_te0);
  LdsTab4 lt = lds_tab4_make(lds_te0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane & 1;
  const int q = wave >> 1, r = wave & 1;       // hash id, gate block of 32
  uint32_t rk[44];
  for (int i = 0; i < 44; i++) rk[i] = c_rk[i];
  uint32_t x0 = (uint32_t)lane * 2654435761u + wave, x1 = x0 * 40503u + 7u;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    uint32_t *xb = xch + (i & 1) * 2048;
    uint32_t o0, o1;
    hash_pair(lt, rk, x0 ^ (uint32_t)q, x1, (uint64_t)i * 64 + q, h, o0, o1);
    xb[q * 512 + r * 128 + lane] = o0;
    xb[q * 512 + r * 128 + 64 + lane] = o1;
    lds_barrier();
    uint32_t a0 = 0, a1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { a0 ^= xb[((k * 3) % (NW / 2)) * 512 + r * 128 + lane]; a1 ^= xb[((k * 3) % (NW / 2)) * 512 + r * 128 + 64 + lane]; }
    x0 = a0; x1 = a1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = x0 ^ x1; }
}
template <int NW> void run() {
  unsigned long long *d; (void)hipMalloc(&d, 64);
  int iters = 2000;
  hipLaunchKernelGGL((lat_kernel<NW>), dim3(1), dim3(NW * 64), 0, 0, d, 10); (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((lat_kernel<NW>), dim3(1), dim3(NW * 64), 0, 0, d, iters); (void)hipDeviceSynchronize();
  unsigned long long h[3]; (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("two-lane hash: %2d waves = %d hashes of 64 gates per level: %7.0f cycles/level %6.3f us\n", NW, NW / 2, (double)h[0] / iters,
         (double)h[1] / iters / 100.0);
  (void)hipFree(d);
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); (void)hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
  uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
  uint32_t *bad; (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(128), 0, 0, bad); (void)hipDeviceSynchronize();
  uint32_t hb = 1; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("two-lane hash vs whole-block hash: %u mismatching columns of 256\n", hb);
  run<2>(); run<4>(); run<8>();
  return 0;
}
