// Experiment (not product code), round 6: can bitsliced AES waves (VALU only, no LDS) run BESIDE the T-table AES waves
// (LDS-lookup bound, about half of the vector issue idle) on the same CU, and what does the CU then hash per second?
//
//   T kernel: the product's four-table AES (gc_device.h LdsTab4, 128 KiB of LDS, one workgroup per CU), NB interleaved
//             blocks per lane, chained.
//   B kernel: bitsliced AES (bsaes.h), 32 blocks per lane in 128 bit planes, 4 waves per workgroup (one per SIMD), no LDS
//             table; with or without the 32 x 32 bit transposes into and out of the plane domain.
// The two kernels are launched on two streams and meet on the CUs (the T kernel's VGPR budget leaves room for one B wave
// per SIMD).  Every wave hashes until a deadline on the 100 MHz wall clock and reports how many blocks it finished; the
// rate is blocks / duration.  Workgroups record their hardware ids, so the report says on how many CUs the two really shared.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o bin/aes_hybrid aes_hybrid.hip && bin/aes_hybrid
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <set>
#include <vector>

#include "../../linreg-mpc_amd/csrc/gc_device.h"
#include "bsaes.h"

using namespace gc;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__constant__ BsKey c_bskey;

struct WaveRep {
    uint64_t blocks;     // per wave (all lanes)
    uint64_t ticks;      // wall-clock ticks (100 MHz) from its first to its last block
    uint64_t t_start;    // absolute wall clock at its first block (one counter for the whole device)
    uint32_t hw_id, xcc_id;
    uint32_t check, pad;
};

__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }

// ------------------------------------------------------------------ T-table waves
template <int TPB, int NB, bool STATIC_LDS>
__device__ __forceinline__ void t_body(WaveRep *rep, uint64_t dur, uint64_t max_blocks = ~0ull) {
    // dynamic LDS: with a static 128 KiB array the compiler knows that only one workgroup fits a CU and pads the kernel's VGPR
    // allocation up to 512 / (waves per SIMD) + 1 -- the T kernel would then own the whole register file
    extern __shared__ uint32_t lds_dyn[];
    __shared__ uint32_t lds_static[STATIC_LDS ? 2 * kLdsTabWords : 1];
    uint32_t *lds_te0 = STATIC_LDS ? lds_static : lds_dyn;
    lds_tab4_fill(lds_te0);
    LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
    const uint64_t t0 = wall_clock64();
    uint64_t t1 = t0, n = 0;
    do {
#pragma unroll 1
        for (int i = 0; i < 4; i++) aes_encrypt_n<NB, LdsTab4>(lt, c_aes.rk, s, c_aes.rk24);
        n += 4 * NB;
        t1 = wall_clock64();
    } while (t1 - t0 < dur && n < max_blocks);
    uint32_t acc = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
    for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) {
        WaveRep r;
        r.blocks = n * 64; r.ticks = t1 - t0; r.t_start = t0; r.hw_id = hw_id(); r.xcc_id = xcc_id(); r.check = acc; r.pad = 0;
        rep[gid >> 6] = r;
    }
}

// ------------------------------------------------------------------ bitsliced waves
// TR = 1: every batch goes natural -> planes -> natural (what a consumer of labels pays); TR = 0: the planes are chained
template <int TPB, int TR, int LDSPAD>
__device__ __forceinline__ void b_body(WaveRep *rep, uint64_t dur) {
    __shared__ uint32_t pad[LDSPAD > 0 ? LDSPAD : 1];   // only to keep a second B workgroup off the CU
    if (LDSPAD > 0 && threadIdx.x == 0) pad[0] = 0;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[128];
#pragma unroll
    for (int i = 0; i < 128; i++) s[i] = gid * 2654435761u + (uint32_t)i * 0x9e3779b9u;
    const uint64_t t0 = wall_clock64();
    uint64_t t1 = t0, n = 0;
    do {
        if (TR) {
            // out of the plane domain and back in, with the whitening key: 8 transposes of 32 x 32 bits
#pragma unroll
            for (int c = 0; c < 4; c++) {
                bs_transpose32(s + 32 * c);
#pragma unroll
                for (int j = 0; j < 32; j++) s[32 * c + j] ^= c_bskey.k0[c];
                bs_transpose32(s + 32 * c);
            }
        }
        bs_encrypt_planes(s, c_bskey);
        n += 32;
        t1 = wall_clock64();
    } while (t1 - t0 < dur);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 128; i++) acc ^= s[i];
    for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) {
        WaveRep r;
        r.blocks = n * 64; r.ticks = t1 - t0; r.t_start = t0; r.hw_id = hw_id(); r.xcc_id = xcc_id(); r.check = acc + (LDSPAD > 0 ? pad[0] : 0); r.pad = 0;
        rep[gid >> 6] = r;
    }
}

// one kernel per shape.  The VGPR allocation is then what the code needs (T: 80 with four blocks per lane, 48-64 with two;
// B: 192) and a B wave fits beside four T waves on every SIMD.
#define T_KERNEL(TPB, NB) \
    __global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(TPB / 256, 8))) t_kernel_##TPB##_##NB(WaveRep *rep, uint64_t dur) { t_body<TPB, NB, false>(rep, dur); } \
    static void launch_t_##TPB##_##NB(WaveRep *rep, uint64_t dur, int grid, hipStream_t st) { \
        static bool once = false; \
        if (!once) { CHK(hipFuncSetAttribute((const void *)t_kernel_##TPB##_##NB, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kLdsTabWords * 4)); once = true; } \
        hipLaunchKernelGGL(t_kernel_##TPB##_##NB, dim3(grid), dim3(TPB), 2 * kLdsTabWords * 4, st, rep, dur); }
#define TS_KERNEL(TPB, NB) \
    __global__ void __launch_bounds__(TPB) ts_kernel_##TPB##_##NB(WaveRep *rep, uint64_t dur) { t_body<TPB, NB, true>(rep, dur); } \
    static void launch_ts_##TPB##_##NB(WaveRep *rep, uint64_t dur, int grid, hipStream_t st) { hipLaunchKernelGGL(ts_kernel_##TPB##_##NB, dim3(grid), dim3(TPB), 0, st, rep, dur); }
// the static-LDS window kernel ended by a block count: timed by events, for the comparison with fixed_work_kernel
__global__ void __launch_bounds__(1024) ts_count_kernel(WaveRep *rep, uint64_t max_blocks) { t_body<1024, 4, true>(rep, ~0ull >> 2, max_blocks); }
#define B_KERNEL(TPB) \
    __global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(TPB / 256, 8))) b_kernel_##TPB(WaveRep *rep, uint64_t dur) { b_body<TPB, 0, 0>(rep, dur); } \
    static void launch_b_##TPB(WaveRep *rep, uint64_t dur, int grid, hipStream_t st) { hipLaunchKernelGGL(b_kernel_##TPB, dim3(grid), dim3(TPB), 0, st, rep, dur); }
T_KERNEL(1024, 4)
T_KERNEL(768, 4)
T_KERNEL(512, 4)
T_KERNEL(1024, 2)
T_KERNEL(768, 2)
T_KERNEL(512, 2)
T_KERNEL(256, 4)
TS_KERNEL(1024, 4)
TS_KERNEL(1024, 2)
TS_KERNEL(768, 4)
TS_KERNEL(768, 2)
TS_KERNEL(1024, 1)
B_KERNEL(256)
B_KERNEL(512)

// the product's micro-kernel (gc_engine.hip gc_aes_bench_kernel) as it is there: a fixed amount of work per lane, timed by
// HIP events around the launch -- to see what the fixed-work / many-workgroups form costs against the fixed-window form above
// MODE 0: as the product's.  1: every wave starts late by (wave index) x `arg` s_sleep units (64 clk each).  2: reads the
// wall clock every four batches (what the fixed-window kernels above do).  3: sleeps (wave & 3) units every `arg` batches.
template <int MODE, int NB>
__global__ void __launch_bounds__(1024)
fixed_work_kernel(uint32_t *out, int blocks_per_lane, int arg, unsigned long long *stamps) {
    const uint64_t t_in = wall_clock64();
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    LdsTab4 lt = lds_tab4_make(lds_te0);
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[NB][4];
    for (int b = 0; b < NB; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE == 1) for (int k = 0; k < wave * arg; k++) __builtin_amdgcn_s_sleep(1);
    uint64_t sink = 0;
    __shared__ uint32_t wg_next;
    if (MODE == 6) { if (threadIdx.x == 0) wg_next = 0; __syncthreads(); }
    if (MODE == 5) {            // rotating user priorities: the SIMD's issue arbiter otherwise favours its oldest wave
        for (int i = 0, k = 0; i < blocks_per_lane; i += NB, k++) {
            if ((k % arg) == 0) {
                switch (((wave >> 2) + k / arg) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            aes_encrypt_n<NB, LdsTab4>(lt, c_aes.rk, s, c_aes.rk24);
        }
    } else if (MODE == 6) {     // the workgroup's work (16 waves x blocks_per_lane) in units of `arg` batches, pulled from a counter
        const uint32_t units = (uint32_t)((blockDim.x >> 6) * (blocks_per_lane / (NB * arg)));
        for (;;) {
            uint32_t u = 0;
            if ((threadIdx.x & 63) == 0) u = atomicAdd(&wg_next, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= units) break;
            for (int k = 0; k < arg; k++) aes_encrypt_n<NB, LdsTab4>(lt, c_aes.rk, s, c_aes.rk24);
        }
    } else if (MODE == 4) {            // the loop of the fixed-window kernels, ended by a count instead of the clock
        int done = 0;
        const uint64_t t0 = wall_clock64();
        uint64_t t1 = t0;
        do {
#pragma unroll 1
            for (int i = 0; i < 4; i++) aes_encrypt_n<NB, LdsTab4>(lt, c_aes.rk, s, c_aes.rk24);
            done += 4 * NB;
            t1 = wall_clock64();
        } while (done < blocks_per_lane && t1 - t0 < (uint64_t)1 << 40);
        sink = t1;
    } else
    for (int i = 0, k = 0; i < blocks_per_lane; i += NB, k++) {
        aes_encrypt_n<NB, LdsTab4>(lt, c_aes.rk, s, c_aes.rk24);
        if (MODE == 2 && (k & 3) == 3) sink += wall_clock64();
        if (MODE == 3 && (k % arg) == arg - 1) { if (wave & 1) __builtin_amdgcn_s_sleep(1); if (wave & 2) __builtin_amdgcn_s_sleep(2); }
    }
    s[0][0] ^= (uint32_t)(sink >> 40);
    uint32_t acc = 0;
    for (int b = 0; b < NB; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
    out[gid] = acc;
    if (stamps && (threadIdx.x & 63) == 0) {
        const unsigned long long t_out = wall_clock64();
        atomicMin(&stamps[0], (unsigned long long)t_in);
        atomicMax(&stamps[1], t_out);
        // how long each wave of a workgroup took (summed over the workgroups): slot 2 + wave
        atomicAdd(&stamps[2 + wave], t_out - t_in);
    }
}

// known-answer path of the bitsliced cipher on the device: 32 blocks per lane through load / encrypt / store
__global__ void __launch_bounds__(64)
b_verify_kernel(const uint32_t *in, uint32_t *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t w[32][4];
    for (int b = 0; b < 32; b++)
        for (int c = 0; c < 4; c++) w[b][c] = in[(lane * 32 + b) * 4 + c];
    uint32_t s[128];
    bs_load(s, w, c_bskey.k0);
    bs_encrypt_planes(s, c_bskey);
    bs_store(s, w);
    for (int b = 0; b < 32; b++)
        for (int c = 0; c < 4; c++) out[(lane * 32 + b) * 4 + c] = w[b][c];
}

// ------------------------------------------------------------------ host
struct Side {
    const char *name;
    void (*launch)(WaveRep *, uint64_t, int, hipStream_t);
    int tpb;
};

static uint32_t cu_key(const WaveRep &r) {
    // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
    return ((r.xcc_id & 0xf) << 16) | (r.hw_id & 0xff00);
}

struct Result { double blocks; int waves; size_t cus; uint64_t first, last; };
static Result summarize(const std::vector<WaveRep> &v, std::set<uint32_t> *cus_out = 0) {
    Result r = {0, (int)v.size(), 0, ~0ull, 0};
    std::set<uint32_t> cus;
    for (auto &w : v) {
        r.blocks += (double)w.blocks;
        cus.insert(cu_key(w));
        if (w.t_start < r.first) r.first = w.t_start;
        if (w.t_start + w.ticks > r.last) r.last = w.t_start + w.ticks;
    }
    r.cus = cus.size();
    if (cus_out) *cus_out = cus;
    return r;
}

int main(int argc, char **argv) {
    double dur_ms = argc > 1 ? atof(argv[1]) : 10.0;
    CHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d MHz; every wave hashes for %.1f ms of wall clock\n", prop.name, ncu, prop.clockRate / 1000, dur_ms);

    // keys
    AesTables t;
    aes_build_tables(t, kFixedKey);
    BsKey bk;
    bs_fold_key(t.rk, bk);
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_bskey), &bk, sizeof(bk)));

    // ---- the bitsliced cipher against the host's table AES
    {
        const int nblk = 64 * 32;
        std::vector<uint32_t> in(nblk * 4), out(nblk * 4);
        srand(7);
        for (auto &x : in) x = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
        uint32_t *di, *dout;
        CHK(hipMalloc(&di, in.size() * 4));
        CHK(hipMalloc(&dout, in.size() * 4));
        CHK(hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(b_verify_kernel, dim3(1), dim3(64), 0, 0, di, dout);
        CHK(hipMemcpy(out.data(), dout, in.size() * 4, hipMemcpyDeviceToHost));
        HostTab ht;
        ht.te0 = t.te0;
        int bad = 0;
        for (int b = 0; b < nblk; b++) {
            uint32_t s[1][4];
            memcpy(s[0], &in[b * 4], 16);
            aes_encrypt_n<1, HostTab>(ht, t.rk, s);
            if (memcmp(s[0], &out[b * 4], 16)) bad++;
        }
        printf("bitsliced AES on the device vs host table AES: %d of %d blocks differ%s\n", bad, nblk, bad ? "  ** WRONG **" : " (exact)");
        if (bad) return 1;
    }

    int wall_khz = 0;
    CHK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall_clock64 rate: %d kHz\n", wall_khz);
    const double kTicksPerMs = (double)wall_khz;
    const uint64_t dur = (uint64_t)(dur_ms * kTicksPerMs);
    const double dur_s = dur_ms * 1e-3;
    WaveRep *rep_t, *rep_b;
    const int kMaxWaves = 4096 * 4;
    CHK(hipMalloc(&rep_t, sizeof(WaveRep) * kMaxWaves));
    CHK(hipMalloc(&rep_b, sizeof(WaveRep) * kMaxWaves));
    hipStream_t st_t, st_b;
    CHK(hipStreamCreateWithFlags(&st_t, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&st_b, hipStreamNonBlocking));

    const Side T[] = {
        {"T 16w nb4", launch_t_1024_4, 1024},
        {"T 12w nb4", launch_t_768_4, 768},
        {"T  8w nb4", launch_t_512_4, 512},
        {"T 16w nb2", launch_t_1024_2, 1024},
        {"T 12w nb2", launch_t_768_2, 768},
        {"T  8w nb2", launch_t_512_2, 512},
        {"T  4w nb4", launch_t_256_4, 256},
        // static 128 KiB LDS array = the product's shape; VGPR allocation padded by the compiler: no room for B beside them
        {"Ts16w nb4", launch_ts_1024_4, 1024},
        {"Ts16w nb2", launch_ts_1024_2, 1024},
        {"Ts12w nb4", launch_ts_768_4, 768},
        {"Ts12w nb2", launch_ts_768_2, 768},
        {"Ts16w nb1", launch_ts_1024_1, 1024},
    };
    const int nTdyn = 7;
    const Side B[] = {
        {"B 4w", launch_b_256, 256},
        {"B 8w", launch_b_512, 512},
    };
    const int nT = sizeof(T) / sizeof(T[0]), nB = sizeof(B) / sizeof(B[0]);

    // rate = all blocks of both kernels / the window from the first wave's start to the last wave's end: two kernels that
    // do not share the CUs run one after the other and gain nothing
    auto run = [&](int ti, int bi, int bgrid_mul) {
        const int tw = ti >= 0 ? ncu * T[ti].tpb / 64 : 0;
        const int bw = bi >= 0 ? ncu * bgrid_mul * B[bi].tpb / 64 : 0;
        CHK(hipMemset(rep_t, 0, sizeof(WaveRep) * kMaxWaves));
        CHK(hipMemset(rep_b, 0, sizeof(WaveRep) * kMaxWaves));
        CHK(hipDeviceSynchronize());
        if (ti >= 0) T[ti].launch(rep_t, dur, ncu, st_t);
        if (bi >= 0) B[bi].launch(rep_b, dur, ncu * bgrid_mul, st_b);
        CHK(hipDeviceSynchronize());
        std::vector<WaveRep> vt(tw), vb(bw);
        if (tw) CHK(hipMemcpy(vt.data(), rep_t, sizeof(WaveRep) * tw, hipMemcpyDeviceToHost));
        if (bw) CHK(hipMemcpy(vb.data(), rep_b, sizeof(WaveRep) * bw, hipMemcpyDeviceToHost));
        std::set<uint32_t> ct, cb;
        Result rt = summarize(vt, &ct), rb = summarize(vb, &cb);
        size_t shared = 0;
        for (auto c : ct) shared += cb.count(c);
        const uint64_t first = rt.first < rb.first ? rt.first : rb.first, last = rt.last > rb.last ? rt.last : rb.last;
        const double win_s = (double)(last - first) / (kTicksPerMs * 1e3);
        printf("%-9s | %-4s x%d | T %.3e  B %.3e  sum %.3e blocks/s over the joint window of %6.2f ms | CUs T %3zu B %3zu both %3zu\n",
               ti >= 0 ? T[ti].name : "-", bi >= 0 ? B[bi].name : "-", bgrid_mul, rt.blocks / win_s, rb.blocks / win_s, (rt.blocks + rb.blocks) / win_s,
               win_s * 1e3, rt.cus, rb.cus, shared);
        fflush(stdout);
        return (rt.blocks + rb.blocks) / win_s;
    };

    {   // fixed work, event-timed (the form of lgc_aes_bench): workgroups x blocks per lane
        uint32_t *fout;
        CHK(hipMalloc(&fout, (size_t)16384 * 1024 * 4));
        hipEvent_t ea, eb;
        CHK(hipEventCreate(&ea)); CHK(hipEventCreate(&eb));
        printf("\n-- the product's micro-kernel, fixed work timed by events (lgc_aes_bench runs 4096 workgroups x 256 blocks per lane)\n");
        unsigned long long *stamps;
        CHK(hipMalloc(&stamps, 18 * 8));
        bool show_waves = false;
        auto timed = [&](const char *name, void (*k)(uint32_t *, int, int, unsigned long long *), int wgs, int bpl, int arg) {
            unsigned long long init[18] = {~0ull, 0ull}, got[18];
            CHK(hipMemcpy(stamps, init, 18 * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, fout, 8, arg, (unsigned long long *)0);
            CHK(hipEventRecord(ea, 0));
            hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, fout, bpl, arg, stamps);
            CHK(hipEventRecord(eb, 0));
            CHK(hipEventSynchronize(eb));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, ea, eb));
            CHK(hipMemcpy(got, stamps, 18 * 8, hipMemcpyDeviceToHost));
            const double win_ms = (double)(got[1] - got[0]) / 1e5;
            printf("%-44s %6d workgroups x %6d blocks per lane: events %8.3f ms  %.3e blocks/s | first wave in .. last wave out %8.3f ms  %.3e\n", name, wgs, bpl, ms,
                   (double)wgs * 1024.0 * bpl / (ms * 1e-3), win_ms, (double)wgs * 1024.0 * bpl / (win_ms * 1e-3));
            if (show_waves) {
                printf("      mean time in the kernel by wave index (ms):");
                for (int w = 0; w < 16; w++) printf(" %.2f", (double)got[2 + w] / 1e5 / wgs);
                printf("\n");
            }
        };
        show_waves = true;
        timed("as the product's (nb4)", fixed_work_kernel<0, 4>, 256, 16384, 0);
        timed("as the product's (nb4)", fixed_work_kernel<0, 4>, 4096, 1024, 0);
        for (int a : {1, 4, 16, 64}) {
            char nm[64]; snprintf(nm, sizeof nm, "nb4, user priority rotates every %d batches", a);
            timed(nm, fixed_work_kernel<5, 4>, 4096, 1024, a);
        }
        for (int a : {1, 4, 16}) {
            char nm[64]; snprintf(nm, sizeof nm, "nb4, units of %d batches pulled by the waves", a);
            timed(nm, fixed_work_kernel<6, 4>, 4096, 1024, a);
        }
        timed("nb4, units of 4 batches pulled by the waves", fixed_work_kernel<6, 4>, 256, 16384, 4);
        show_waves = false;
        timed("as the product's (nb4)", fixed_work_kernel<0, 4>, 256, 4096, 0);
        timed("as the product's (nb4)", fixed_work_kernel<0, 4>, 256, 16384, 0);
        timed("as the product's (nb4)", fixed_work_kernel<0, 4>, 4096, 1024, 0);
        timed("nb2", fixed_work_kernel<0, 2>, 256, 16384, 0);
        timed("nb2", fixed_work_kernel<0, 2>, 4096, 1024, 0);
        timed("nb1", fixed_work_kernel<0, 1>, 4096, 1024, 0);
        for (int a : {1, 2, 4, 8}) {
            char nm[64]; snprintf(nm, sizeof nm, "nb4, wave w starts w x %d sleep units late", a);
            timed(nm, fixed_work_kernel<1, 4>, 4096, 1024, a);
        }
        timed("nb4, wall clock read every 4 batches", fixed_work_kernel<2, 4>, 4096, 1024, 0);
        timed("nb4, wall clock read every 4 batches", fixed_work_kernel<2, 4>, 256, 16384, 0);
        for (int a : {1, 4, 16}) {
            char nm[64]; snprintf(nm, sizeof nm, "nb4, (wave & 3) sleep units every %d batches", a);
            timed(nm, fixed_work_kernel<3, 4>, 4096, 1024, a);
        }
        timed("nb2, wall clock read every 4 batches", fixed_work_kernel<2, 2>, 4096, 1024, 0);
        {
            CHK(hipEventRecord(ea, 0));
            hipLaunchKernelGGL(ts_count_kernel, dim3(256), dim3(1024), 0, 0, rep_t, (uint64_t)16384);
            CHK(hipEventRecord(eb, 0));
            CHK(hipEventSynchronize(eb));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, ea, eb));
            printf("the static-LDS WINDOW kernel (Ts16w nb4) ended by a count of 16384 blocks per lane, 256 workgroups: events %8.3f ms  %.3e blocks/s\n", ms, 256.0 * 1024 * 16384 / (ms * 1e-3));
        }
        timed("nb4, the window kernels' loop, count-ended", fixed_work_kernel<4, 4>, 256, 16384, 0);
        timed("nb4, the window kernels' loop, count-ended", fixed_work_kernel<4, 4>, 4096, 1024, 0);
        timed("nb1, the window kernels' loop, count-ended", fixed_work_kernel<4, 1>, 4096, 1024, 0);
        CHK(hipFree(fout));
    }
    if (argc > 2) return 0;     // only the fixed-work table
    run(0, -1, 1);   // warm-up (code objects, clocks)
    printf("\n-- T-table alone (one workgroup per CU)\n");
    double base = 0;
    double best_t = 0;
    int best_ti = 0;
    for (int rep = 0; rep < 2; rep++)
        for (int i = 0; i < nT; i++) {
            double r = run(i, -1, 1);
            if (i == 7 && r > base) base = r;
            if (r > best_t) { best_t = r; best_ti = i; }
        }
    printf("\n-- bitsliced alone (workgroups per CU: 1, 2)\n");
    for (int i = 0; i < nB; i++) { run(-1, i, 1); run(-1, i, 2); }
    printf("\n-- together\n");
    double best = 0;
    int bt = -1, bb = -1, bm = 1;
    for (int i = 0; i < nTdyn; i++)
        for (int j = 0; j < nB; j++)
            for (int m = 1; m <= 2; m++) {
                if (j == 1 && m == 2) continue;
                double r = run(i, j, m);
                if (r > best) { best = r; bt = i; bb = j; bm = m; }
            }
    printf("\nproduct's micro-kernel shape (Ts16w nb4): %.3e blocks/s; best T-table-only shape (%s): %.3e; best mix: %s + %s x%d: %.3e\n"
           "  = %.3f x the product's shape, %.3f x the best T-table-only shape; LDS lookup roof (160 ds_read_b32 x 2 clk per wave-block): %.3e\n",
           base, T[best_ti].name, best_t, T[bt].name, B[bb].name, bm, best, best / base, best / best_t, (double)ncu * prop.clockRate * 1e3 * 64.0 / 320.0);
    return 0;
}
