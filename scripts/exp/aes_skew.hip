// experiment: two AES blocks per lane, skewed by half a round (one block's lookups are in flight
// while the other block's XORs run) vs the plain round-by-round interleave (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

__device__ __forceinline__ void issue(const LdsTab4 &t, const uint32_t s[4], uint32_t v[16]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[4 * j + 0] = t.lkt(0, s[j], 0);
        v[4 * j + 1] = t.lkt(1, s[(j + 1) & 3], 1);
        v[4 * j + 2] = t.lkt(2, s[(j + 2) & 3], 2);
        v[4 * j + 3] = t.lkt(3, s[(j + 3) & 3], 3);
    }
}
__device__ __forceinline__ void mix(const uint32_t v[16], const uint32_t *rk, int rnd, uint32_t s[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = xor3(xor3(v[4 * j], v[4 * j + 1], rk[4 * rnd + j]), v[4 * j + 2], v[4 * j + 3]);
}
__device__ __forceinline__ void issue_last(const LdsTab4 &t, const uint32_t s[4], uint32_t v[16]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[4 * j + 0] = t.lkt(0, s[j], 0);
        v[4 * j + 1] = t.lkt(0, s[(j + 1) & 3], 1);
        v[4 * j + 2] = t.lkt(0, s[(j + 2) & 3], 2);
        v[4 * j + 3] = t.lkt(0, s[(j + 3) & 3], 3);
    }
}
__device__ __forceinline__ void mix_last(const uint32_t v[16], const uint32_t *rk, uint32_t s[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = xor3(last_lo(v[4 * j + 1], v[4 * j]), last_hi(v[4 * j + 3], v[4 * j + 2]), rk[40 + j]);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// A leads B by half a round
__device__ __forceinline__ void aes2_skew(const LdsTab4 &t, const uint32_t *rk, uint32_t a[4], uint32_t b[4]) {
    uint32_t va[16], vb[16];
#pragma unroll
    for (int j = 0; j < 4; j++) { a[j] ^= rk[j]; b[j] ^= rk[j]; }
    issue(t, a, va); FENCE();
    issue(t, b, vb); FENCE();
#pragma unroll
    for (int r = 1; r < 9; r++) {
        mix(va, rk, r, a); issue(t, a, va); FENCE();      // B's lookups are in flight
        mix(vb, rk, r, b); issue(t, b, vb); FENCE();      // A's lookups are in flight
    }
    mix(va, rk, 9, a); issue_last(t, a, va); FENCE();
    mix(vb, rk, 9, b); issue_last(t, b, vb); FENCE();
    mix_last(va, rk, a); FENCE();
    mix_last(vb, rk, b);
}

template <int MODE>   // 0: aes_encrypt_n<2>; 1: skewed pair; 2: aes_encrypt_n<4>
__global__ void __launch_bounds__(1024) k_bench(uint32_t *out, int blocks_per_lane) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    LdsTab4 lt = lds_tab4_make(lds_te0);
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[4][4];
    for (int b = 0; b < 4; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
    for (int i = 0; i < blocks_per_lane; i += 4) {
        if (MODE == 0) { aes_encrypt_n<2, LdsTab4>(lt, c_rk, s, c_rk24); aes_encrypt_n<2, LdsTab4>(lt, c_rk, s + 2, c_rk24); }
        else if (MODE == 1) { aes2_skew(lt, c_rk, s[0], s[1]); aes2_skew(lt, c_rk, s[2], s[3]); }
        else aes_encrypt_n<4, LdsTab4>(lt, c_rk, s, c_rk24);
    }
    uint32_t acc = 0;
    for (int b = 0; b < 4; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
    out[gid] = acc;
}
template <int MODE> void run(const char *name) {
    int nblk = 4096, bpl = 256;
    uint32_t *out; hipMalloc(&out, (size_t)nblk * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(nblk), dim3(1024), 0, 0, out, 4);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(nblk), dim3(1024), 0, 0, out, bpl);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint32_t h[64]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
    uint32_t c = 0; for (int i = 0; i < 64; i++) c = c * 31 + h[i];
    printf("%-44s %.3e AES/s  check %08x\n", name, (double)nblk * 1024 * bpl / (ms * 1e-3), c);
    hipFree(out);
}
int main() {
    AesTables t; aes_build_tables(t, kFixedKey);
    uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
    hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
    hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
    for (int rep = 0; rep < 2; rep++) {
        run<0>("2 blocks interleaved round by round");
        run<1>("2 blocks skewed by half a round");
        run<2>("4 blocks interleaved round by round");
    }
    return 0;
}
