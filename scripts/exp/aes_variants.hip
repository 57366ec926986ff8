// experiment: AES formulations on gfx950 (not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../linreg-mpc_amd/csrc/gc_aes.h"
using namespace gc;
__constant__ uint32_t k_rk[44];
__constant__ uint32_t k_te0[256];

// ---------- V0: current (32x replicated, compiler code)
struct Tab32 { const uint32_t* base; __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i << 5]; } };
template<int N>
__global__ void __launch_bounds__(256) v0_kernel(uint32_t* out, int iters) {
  __shared__ uint32_t lds[256*32];
  for (int i = threadIdx.x; i < 256*32; i += blockDim.x) lds[i] = k_te0[i >> 5];
  __syncthreads();
  Tab32 t; t.base = lds + (threadIdx.x & 31);
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s[N][4];
  for (int b = 0; b < N; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
  for (int i = 0; i < iters; i++) aes_encrypt_n<N, Tab32>(t, k_rk, s);
  uint32_t acc = 0; for (int b = 0; b < N; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
  out[gid] = acc;
}

// ---------- V1: 64x replicated (256 B per entry), v_perm address formation, perm last round
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t xor3s(uint32_t a, uint32_t b, uint32_t k) {
  return __builtin_amdgcn_bitop3_b32(a, b, k, 0x96); }
template<int N, int TPB>
__global__ void __launch_bounds__(TPB) v1_kernel(uint32_t* out, int iters) {
  __shared__ uint32_t lds[256*64];
  for (int i = threadIdx.x; i < 256*64; i += blockDim.x) lds[i] = k_te0[i >> 6];
  __syncthreads();
  const uint32_t lane4 = (threadIdx.x & 63) << 2;
  const char* L = (const char*)lds;
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s[N][4];
  for (int b = 0; b < N; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int b = 0; b < N; b++) { s[b][0] ^= k_rk[0]; s[b][1] ^= k_rk[1]; s[b][2] ^= k_rk[2]; s[b][3] ^= k_rk[3]; }
#pragma unroll
    for (int r = 1; r < 10; r++) {
      uint32_t v[N][16];
#pragma unroll
      for (int b = 0; b < N; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          v[b][4*j+0] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][j], lane4, 0x0c0c0400));
          v[b][4*j+1] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+1)&3], lane4, 0x0c0c0500));
          v[b][4*j+2] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+2)&3], lane4, 0x0c0c0600));
          v[b][4*j+3] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+3)&3], lane4, 0x0c0c0700));
        }
#pragma unroll
      for (int b = 0; b < N; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t a = xor3s(v[b][4*j], __builtin_amdgcn_alignbit(v[b][4*j+2], v[b][4*j+2], 16), k_rk[4*r+j]);
          s[b][j] = xor3(a, __builtin_amdgcn_alignbit(v[b][4*j+1], v[b][4*j+1], 24), __builtin_amdgcn_alignbit(v[b][4*j+3], v[b][4*j+3], 8));
        }
    }
    {
      uint32_t v[N][16];
#pragma unroll
      for (int b = 0; b < N; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          v[b][4*j+0] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][j], lane4, 0x0c0c0400));
          v[b][4*j+1] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+1)&3], lane4, 0x0c0c0500));
          v[b][4*j+2] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+2)&3], lane4, 0x0c0c0600));
          v[b][4*j+3] = *(const uint32_t*)(L + __builtin_amdgcn_perm(s[b][(j+3)&3], lane4, 0x0c0c0700));
        }
#pragma unroll
      for (int b = 0; b < N; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          // Te0 = (2s, s, s, 3s): S[x] sits in bytes 1 and 2
          uint32_t lo = __builtin_amdgcn_perm(v[b][4*j+1], v[b][4*j+0], 0x0c0c0501);   // byte0 <- S1.b1, byte1 <- S0.b1
          uint32_t hi = __builtin_amdgcn_perm(v[b][4*j+3], v[b][4*j+2], 0x05020c0c);   // byte2 <- S1.b2, byte3 <- S0.b1
          s[b][j] = xor3s(lo, hi, k_rk[40+j]);
        }
    }
  }
  uint32_t acc = 0; for (int b = 0; b < N; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
  out[gid] = acc;
}

template<class F> double timeit(F f, uint32_t* d_out, size_t nthreads, int iters, int N, uint32_t* chk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(4);
  hipEventRecord(a); f(iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<uint32_t> h(nthreads); hipMemcpy(h.data(), d_out, nthreads*4, hipMemcpyDeviceToHost);
  uint32_t c = 0; for (auto x : h) c ^= x; *chk = c;
  return (double)nthreads * iters * N / (ms * 1e-3);
}
int main() {
  AesTables t; aes_build_tables(t, kFixedKey);
  hipMemcpyToSymbol(HIP_SYMBOL(k_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(k_te0), t.te0, sizeof(t.te0));
  const int waves = 65536; size_t nthreads = (size_t)waves * 64; int iters = 64;
  uint32_t* d_out; hipMalloc(&d_out, nthreads * 4);
  uint32_t c;
  double r;
  r = timeit([&](int it){ hipLaunchKernelGGL((v0_kernel<4>), dim3(nthreads/256), dim3(256), 0, 0, d_out, it); }, d_out, nthreads, iters, 4, &c);
  printf("V0 N=4 tpb256: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v0_kernel<2>), dim3(nthreads/256), dim3(256), 0, 0, d_out, it); }, d_out, nthreads, iters, 2, &c);
  printf("V0 N=2 tpb256: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v1_kernel<4,512>), dim3(nthreads/512), dim3(512), 0, 0, d_out, it); }, d_out, nthreads, iters, 4, &c);
  printf("V1 N=4 tpb512: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v1_kernel<2,512>), dim3(nthreads/512), dim3(512), 0, 0, d_out, it); }, d_out, nthreads, iters, 2, &c);
  printf("V1 N=2 tpb512: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v1_kernel<4,1024>), dim3(nthreads/1024), dim3(1024), 0, 0, d_out, it); }, d_out, nthreads, iters, 4, &c);
  printf("V1 N=4 tpb1024: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v1_kernel<1,512>), dim3(nthreads/512), dim3(512), 0, 0, d_out, it); }, d_out, nthreads, iters, 1, &c);
  printf("V1 N=1 tpb512: %.3e AES/s chk %08x\n", r, c);
  r = timeit([&](int it){ hipLaunchKernelGGL((v1_kernel<8,512>), dim3(nthreads/512), dim3(512), 0, 0, d_out, it); }, d_out, nthreads, iters, 8, &c);
  printf("V1 N=8 tpb512: %.3e AES/s chk %08x\n", r, c);
  return 0;
}
