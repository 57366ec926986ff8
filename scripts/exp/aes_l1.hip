// experiment: send a fraction of the T-table lookups through the vector-memory path (L1-resident
// 4 KiB table, buffer_load) instead of LDS, to relieve the LDS lookup bound (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../linreg-mpc_amd/csrc/gc_device.h"
using namespace gc;

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: all LDS; 1: table 3 via buffer_load in every round; 2: table 3 via buffer_load in odd rounds
struct MixTab {
    static const bool kTwoTables = false;
    static const bool kFourTables = true;
    LdsTab4 l;
    __amdgpu_buffer_rsrc_t rsrc;   // buffer descriptor of the 4 x 256-entry global table
    int rnd;
    __device__ __forceinline__ uint32_t lk(uint32_t word, int k) const { return l.lkt(0, word, k); }
    __device__ __forceinline__ uint32_t lk2(uint32_t word, int k) const { return l.lkt(2, word, k); }
    __device__ __forceinline__ uint32_t lkt(int t, uint32_t word, int k) const {
        if (MODE != 0 && t == 3) {
            uint32_t off = ((word >> (8 * k)) & 0xffu) << 2;
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off + 3 * 1024), 0, 0);
        }
        return l.lkt(t, word, k);
    }
};

template <int MODE>
__global__ void __launch_bounds__(1024) k_bench(uint32_t *out, int blocks_per_lane, const uint32_t *gtab) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    MixTab<MODE> mt;
    mt.l = lds_tab4_make(lds_te0);
    mt.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gtab, 0, 4096, 0x00020000);
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[4][4];
    for (int b = 0; b < 4; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
    for (int i = 0; i < blocks_per_lane; i += 4) aes_encrypt_n<4, MixTab<MODE>>(mt, c_rk, s, c_rk24);
    uint32_t acc = 0;
    for (int b = 0; b < 4; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
    out[gid] = acc;
}

template <int MODE> void run(const char *name, const uint32_t *gtab) {
    int nblk = 4096, bpl = 256;
    uint32_t *out; hipMalloc(&out, (size_t)nblk * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(nblk), dim3(1024), 0, 0, out, 4, gtab);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(nblk), dim3(1024), 0, 0, out, bpl, gtab);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint32_t h[64]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
    uint32_t c = 0; for (int i = 0; i < 64; i++) c = c * 31 + h[i];
    printf("%-44s %.3e AES/s  check %08x  (%s)\n", name, (double)nblk * 1024 * bpl / (ms * 1e-3), c, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}
int main() {
    AesTables t; aes_build_tables(t, kFixedKey);
    uint32_t rk24[44]; for (int i = 0; i < 44; i++) rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8);
    hipMemcpyToSymbol(HIP_SYMBOL(c_rk), t.rk, sizeof(t.rk)); hipMemcpyToSymbol(HIP_SYMBOL(c_te0), t.te0, sizeof(t.te0));
    hipMemcpyToSymbol(HIP_SYMBOL(c_rk24), rk24, sizeof(rk24));
    uint32_t host[1024];
    for (int tt = 0; tt < 4; tt++) for (int x = 0; x < 256; x++) { uint32_t v = t.te0[x]; int r = 8 * tt; host[tt * 256 + x] = r ? ((v << r) | (v >> (32 - r))) : v; }
    uint32_t *gtab; hipMalloc(&gtab, 4096); hipMemcpy(gtab, host, 4096, hipMemcpyHostToDevice);
    run<0>("all 160 lookups in LDS", gtab);
    run<1>("table 3 (36 of 160) through buffer_load", gtab);
    run<0>("all 160 lookups in LDS", gtab);
    return 0;
}
