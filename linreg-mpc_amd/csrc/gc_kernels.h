// gc_kernels.h -- the kernel dispatch of a launch: included by gc_kern.hip only, which is compiled once per (role, kernel
// family, gate hash) (csrc/Makefile) and exports the pieces as plain functions (declared in gc_launch.h).  Everything
// else in the library calls those.
#pragma once
#include <hip/hip_runtime.h>

#include "gc_device.h"
#include "gc_split.h"
#include "gc_launch.h"

namespace gc {

static inline hipError_t gc_launch_tabfill_impl(const Launch &L, const Lbl *stash, Lbl *tab, Lbl R, hipStream_t st) {
    const uint64_t per = kTpbTabfill / 64;
    uint64_t blocks = (L.steps + per - 1) / per;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((gc_tabfill_kernel<kTpbTabfill>), dim3((unsigned)blocks), dim3(kTpbTabfill), 0, st, stash, tab,
                       (uint32_t)L.steps, L.step0, R);
    return hipGetLastError();
}

// the record kernel of a launch in mode `m` (garbler in a critical-path mode: `tab` is the stash)
// PART: which kernels this translation unit instantiates (gc_kern.hip): 0 = MAC, 1 = generic one wave per record (wide), 2 = column-split, 3 = generic 4 waves per record
// HK: the gate hash (gc_aes.h): 0 = fixed-key AES (tables in LDS), 1 = Chaskey-12 permutation (no tables; the
// column-split and critical-path kernels hash in their own AES layouts and exist for HK = 0 only)
template <bool G, int PART, int HK = 0>
static inline hipError_t gc_launch_records_impl(LaunchMode m, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w,
                                    int p, hipStream_t st) {
    switch (m) {
    case LM_NONE:
        return hipSuccess;
    case LM_MAC: if constexpr (PART == 0) {
        constexpr int TPB = HK == 1 ? (G ? kTpbMacGH : kTpbMacEH) : (G ? kTpbMacG : kTpbMacE);       // upper bound (register budget of the kernel)
        const unsigned per = gc_mac_waves(L.nrec, G ? GC_MAC_ADAPT_LO_G : GC_MAC_ADAPT_LO_E, TPB / 64);
#if GC_MAC_TAIL_SPLIT
        // One workgroup per CU, every record the same length: the launch runs in rounds of (CUs x waves) records and
        // a partly filled last round costs a whole one.  The records beyond the last full round therefore go into a
        // second launch of ONE workgroup per CU with just enough waves: an LDS-bound workgroup of w waves takes
        // about w / 16 of the time of a full one, so the tail costs its share instead of a round.
        const uint32_t round = gc_num_cus() * per, full = L.nrec / round * round, rest = L.nrec - full;
        if (!GC_MAC_ADAPT && full && rest && rest < round - round / 8) {
            unsigned wv = (rest + gc_num_cus() - 1) / gc_num_cus();
            hipLaunchKernelGGL((gc_mac_kernel<G, TPB, HK>), dim3(full / per), dim3(per * 64), 0, st, recs + L.first_rec, full, words,
                               tab, L.step0, R, w, p);
            hipLaunchKernelGGL((gc_mac_kernel<G, TPB, HK>), dim3((rest + wv - 1) / wv), dim3(wv * 64), 0, st, recs + L.first_rec + full, rest,
                               words, tab, L.step0, R, w, p);
            return hipGetLastError();
        }
#endif
        unsigned wgs = (L.nrec + per - 1) / per;
#if GC_MAC_PERSIST      /* one workgroup per CU: the waves walk the records themselves (gc_device.h) */
        if (G && wgs > gc_num_cus() && wgs <= GC_MAC_PERSIST_MAX_ROUNDS * gc_num_cus()) wgs = gc_num_cus();
#endif
        hipLaunchKernelGGL((gc_mac_kernel<G, TPB, HK>), dim3(wgs), dim3(per * 64), 0, st, recs + L.first_rec, L.nrec, words, tab,
                           L.step0, R, w, p);
    } break;
    case LM_MACK: if constexpr (PART == 0) {
        constexpr int TPB = HK == 1 ? (G ? kTpbMackGH : kTpbMackEH) : (G ? kTpbMackG : kTpbMackE);
        // HK = 1: no table image limits the workgroups of a CU; kMackPadH1 bytes of unused dynamic LDS per workgroup do (see gc_launch.h)
        unsigned per = TPB / 64;
#if GC_MACK_ADAPT
        // One workgroup per CU, every record the same length: the launch runs in rounds of (CUs x waves) records.  An
        // LDS-bound workgroup of w waves takes about w / 16 of the time of a full one: with fewer waves per workgroup the SAME
        // number of rounds costs less (5 000 pairs: 2 rounds of 12 waves instead of 16 + 3.5).  Only for launches of at most
        // GC_MACK_ADAPT_MAX_ROUNDS rounds (gc_launch.h has the measurements), and HK = 0 only.
        if (HK == 0) per = gc_mack_waves(L.nrec, G ? GC_MACK_ADAPT_LO_G : GC_MACK_ADAPT_LO_E, TPB / 64);
#endif
        hipLaunchKernelGGL((gc_mack_kernel<G, TPB, HK>), dim3((L.nrec + per - 1) / per), dim3(per * 64), HK == 1 ? kMackPadH1 : 0, st,
                           recs + L.first_rec, L.nrec, words, tab, L.step0, R, w, p);
    } break;
    case LM_WIDE: if constexpr (PART == 1) {
        // records (waves) per workgroup: as few as keep the launch within one workgroup per CU, at most TPB / 64 --
        // a launch of 800 dividers runs as 200 workgroups of 4 waves, one round, instead of 67 CUs with 12 waves each
        unsigned per = (L.nrec + gc_num_cus() - 1) / gc_num_cus();
        if (per > (unsigned)kTpbWide / 64) per = gc_wide_waves(L.nrec, kTpbWide / 64);
        if (!GC_WIDE_ADAPT) per = kTpbWide / 64;
        hipLaunchKernelGGL((gc_exec_kernel<G, false, HK == 1 ? 0 : 4, kTpbWide>), dim3((L.nrec + per - 1) / per), dim3(per * 64), 0, st,
                           recs + L.first_rec, L.nrec, words, tab, dec, L.step0, R, w, p);
    } break;
    case LM_SPLIT:
        if constexpr (PART == 2 && HK == 0)
            hipLaunchKernelGGL((gc_split_kernel<G>), dim3(L.nrec), dim3(1024), 0, st, recs + L.first_rec, L.nrec, words, tab, dec,
                               L.step0, R, w, p);
        break;
    case LM_QUAD4:
#if GC_QUAD4          /* (PART 3) the 4-wave kernels with the four-table image (garbler: critical-path garbling): what ran these launches
                         before the column-split kernel; with 0 (default, 45 s less to compile) a role whose split kernel is
                         switched off runs them in the two-table 4-wave kernel below */
        if constexpr (PART == 3 && HK == 0)
            hipLaunchKernelGGL((gc_exec_kernel<G, true, 4, 256, G && GC_CRIT>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec,
                               L.nrec, words, tab, dec, L.step0, R, w, p);
        break;
#endif
    case LM_QUAD2:
        if constexpr (PART == 3)
            hipLaunchKernelGGL((gc_exec_kernel<G, true, HK == 1 ? 0 : 2, 256>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec, L.nrec, words,
                               tab, dec, L.step0, R, w, p);
        break;
    }
    return hipGetLastError();
}

}  // namespace gc
