// gc_kern.hip -- one translation unit per (role, kernel family): compiled with -DGC_KERN_G=0|1 (evaluator | garbler) and
// -DGC_KERN_PART=0|1|2|3 (MAC | generic, one wave per record | column-split | generic, 4 waves per record; the garbler's
// part 2 also holds the table pass of critical-path garbling).  Eight units build in parallel instead of one that takes minutes.
#include "gc_kernels.h"

#define GC_CAT3_(a, b, c) a##b##_##c
#define GC_CAT3(a, b, c) GC_CAT3_(a, b, c)
#if GC_KERN_G
#define GC_ROLE_TAG g
#else
#define GC_ROLE_TAG e
#endif

namespace gc {
hipError_t GC_CAT3(gc_launch_records_, GC_ROLE_TAG, GC_KERN_PART)(LaunchMode m, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec,
                                                                   Lbl *tab, Lbl R, int w, int p, hipStream_t st) {
    return gc_launch_records_impl<GC_KERN_G != 0, GC_KERN_PART>(m, recs, L, words, dec, tab, R, w, p, st);
}
// loads this translation unit's code object (first launch of any of its kernels does: ~5-10 ms) ahead of the first real launch
__global__ void GC_CAT3(gc_kern_touch_kernel_, GC_ROLE_TAG, GC_KERN_PART)() {}
hipError_t GC_CAT3(gc_kern_touch_, GC_ROLE_TAG, GC_KERN_PART)(hipStream_t st) {
    hipLaunchKernelGGL(GC_CAT3(gc_kern_touch_kernel_, GC_ROLE_TAG, GC_KERN_PART), dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
#if GC_KERN_G && GC_KERN_PART == 2
hipError_t gc_launch_tabfill(const Launch &L, const Lbl *stash, Lbl *tab, Lbl R, hipStream_t st) {
    return gc_launch_tabfill_impl(L, stash, tab, R, st);
}
#endif
}  // namespace gc
#if GC_SPLIT_TRACE && GC_KERN_PART == 2
// profiling builds only (-DGC_SPLIT_TRACE=1): the stamps of gc_split.h, and a reset
#if GC_KERN_G
extern "C" int lgc_dbg_split_trace_g(uint64_t *out, uint32_t *n, int reset) {
#else
extern "C" int lgc_dbg_split_trace_e(uint64_t *out, uint32_t *n, int reset) {
#endif
    if (reset) { uint32_t z[2] = {0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(gc::g_split_trace_n), z, sizeof z); }
    hipError_t e = hipMemcpyFromSymbol(n, HIP_SYMBOL(gc::g_split_trace_n), 8);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(gc::g_split_trace), 2 * 8192 * 8);
    return (int)e;
}
#endif
#if GC_MAC_WAVE_TRACE && GC_KERN_PART == 0
// profiling builds only (-DGC_MAC_WAVE_TRACE=1): mean time a wave of a Karatsuba MAC workgroup spends in the kernel, by wave index
#if GC_KERN_G
extern "C" int lgc_dbg_mac_wave_ticks_g(uint64_t *ticks, uint64_t *count, int reset) {
#else
extern "C" int lgc_dbg_mac_wave_ticks_e(uint64_t *ticks, uint64_t *count, int reset) {
#endif
    if (reset) {
        unsigned long long z[32] = {0};
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(gc::g_mac_wave_ticks), z, sizeof z);
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(gc::g_mac_wave_count), z, sizeof z);
        return (int)e;
    }
    hipError_t e = hipMemcpyFromSymbol(ticks, HIP_SYMBOL(gc::g_mac_wave_ticks), 32 * 8);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(count, HIP_SYMBOL(gc::g_mac_wave_count), 32 * 8);
    return (int)e;
}
#endif
