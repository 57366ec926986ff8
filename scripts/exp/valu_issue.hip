// valu_issue -- issue rate of the integer vector instructions the gate hashes are made of, on gfx950 (VERDICT r3 item 9 /
// next-round item 6): dependent and independent chains of v_add_u32, v_xor_b32, v_alignbit_b32, v_perm_b32 and
// v_bitop3_b32 (and v_fma_f32 as the reference point of MI355X_MICROARCH.md:53-54) at 1, 2, 4 and 8 waves per SIMD.
// Every wave times its own loop with s_memtime (shader clock); reported: cycles per wave-instruction as ONE wave sees it,
// and wave-instructions per cycle per SIMD (= waves x instructions / cycles).
// Build: hipcc --offload-arch=gfx950 -O2 -o bin/valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// OP: 0 add, 1 xor, 2 alignbit, 3 perm, 4 bitop3, 5 fma_f32.  DEP: one chain; else eight independent chains.
template <int OP, bool DEP>
__global__ void __launch_bounds__(256) k(unsigned long long *out, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned y = seed | 1u, z = seed * 0x9e3779b9u + 12345u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define ONE(r) \
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(y)); \
        else if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(y)); \
        else if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r) : "v"(y)); \
        else if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(y), "v"(z)); \
        else if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(y), "v"(z)); \
        else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(y), "v"(z)); \
        else if (OP == 6) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(r) : "v"(y)); \
        else if (OP == 7) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(y), "v"(z)); \
        else if (OP == 8) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r)); \
        else if (OP == 9) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r) : "v"(y)); \
        else if (OP == 10) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r) : "v"(y)); \
        else if (OP == 11) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(y), "v"(z)); \
        else if (OP == 12) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(r)); \
        else asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(y), "v"(r));
        if (DEP) { REP64(ONE(a0)) }
        else { REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7)) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) + (s == 0x12345u ? 1 : 0);
}

template <int OP, bool DEP>
static void run(const char *name, int cus) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps, waves = blocks * 4;                    // blocks of 4 waves: one per SIMD
        unsigned long long *d = 0;
        hipMalloc(&d, waves * 8);
        hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);          // warm-up
        hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
        hipFree(d);
        std::sort(h.begin(), h.end());
        double med = (double)h[waves / 2], n = 64.0 * iters;
        printf("%-10s %-4s waves/SIMD %d : %6.2f cycles per wave-instruction (median wave; min %.2f) -> %5.3f wave-instr/clk/SIMD = %5.1f lanes/clk/SIMD\n",
               name, DEP ? "dep" : "ind", wps, med / n, (double)h[0] / n, wps * n / med, 64.0 * wps * n / med);
    }
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d kHz; __builtin_readcyclecounter = s_memtime\n", p.name, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    run<0, true>("v_add_u32", cus);   run<0, false>("v_add_u32", cus);
    run<1, true>("v_xor_b32", cus);   run<1, false>("v_xor_b32", cus);
    run<2, true>("v_alignbit", cus);  run<2, false>("v_alignbit", cus);
    run<3, true>("v_perm_b32", cus);  run<3, false>("v_perm_b32", cus);
    run<4, true>("v_bitop3", cus);    run<4, false>("v_bitop3", cus);
    run<5, true>("v_fma_f32", cus);   run<5, false>("v_fma_f32", cus);
    run<6, true>("mov_sdwa", cus);    run<6, false>("mov_sdwa", cus);          // byte insert, other bytes preserved
    run<7, false>("v_and_or", cus);
    run<8, false>("v_bfe_u32", cus);
    run<9, false>("v_lshl_or", cus);
    run<10, false>("alignbyte", cus);
    run<11, false>("v_bfi_b32", cus);
    run<12, true>("mov_dpp", cus);    run<12, false>("mov_dpp", cus);
    run<13, false>("lshl_sdwa", cus);
    return 0;
}
