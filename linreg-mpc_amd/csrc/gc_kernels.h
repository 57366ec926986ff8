// gc_kernels.h -- the kernel dispatch of a launch: included by gc_kern.hip only, which is compiled once per (role, kernel
// family) (csrc/Makefile) and exports the pieces as plain functions (declared in gc_launch.h).  Everything else in the
// library calls those.
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
// PART: which kernels this translation unit instantiates (gc_kern.hip): 0 = MAC, 1 = generic one wave per record (wide),
// 2 = column-split, 3 = generic 4 waves per record
template <bool G, int PART>
static inline hipError_t gc_launch_records_impl(LaunchMode m, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w,
                                    int p, hipStream_t st) {
    switch (m) {
    case LM_NONE:
        return hipSuccess;
    case LM_MAC: if constexpr (PART == 0) {
        constexpr int TPB = G ? kTpbMacG : kTpbMacE;       // upper bound (register budget of the kernel)
        const unsigned per = gc_mac_waves(L.nrec, kMacAdaptLo, TPB / 64);
        const unsigned per_wg = gc_mac_per_wg(L.nrec, per);
        hipLaunchKernelGGL((gc_mac_kernel<G, TPB>), dim3((L.nrec + per_wg - 1) / per_wg), dim3(per * 64), 0, st, recs + L.first_rec, L.nrec, per_wg,
                           words, tab, L.step0, R, w, p);
    } break;
    case LM_MACK: if constexpr (PART == 0) {
        constexpr int TPB = G ? kTpbMackG : kTpbMackE;
        // One workgroup per CU, every record the same length: the launch runs in rounds of (CUs x waves) records.  An
        // LDS-bound workgroup of w waves takes about w / 16 of the time of a full one: with fewer waves per workgroup the SAME
        // number of rounds costs less (5 000 pairs: 2 rounds of 12 waves instead of 16 + 3.5).  Only for launches of at most
        // kMackAdaptMaxRounds rounds (gc_launch.h has the measurements).
        const unsigned per = gc_mack_waves(L.nrec, kMacAdaptLo, TPB / 64);
        const unsigned per_wg = gc_mac_per_wg(L.nrec, per);
        hipLaunchKernelGGL((gc_mack_kernel<G, TPB>), dim3((L.nrec + per_wg - 1) / per_wg), dim3(per * 64), 0, st,
                           recs + L.first_rec, L.nrec, per_wg, words, tab, L.step0, R, w, p);
    } break;
    case LM_WIDE: if constexpr (PART == 1) {
        // records (waves) per workgroup: as few as keep the launch within one workgroup per CU, at most TPB / 64 --
        // a launch of 800 dividers runs as 200 workgroups of 4 waves, one round, instead of 67 CUs with 12 waves each
        unsigned per = (L.nrec + gc_num_cus() - 1) / gc_num_cus();
        if (per > (unsigned)kTpbWide / 64) per = kTpbWide / 64;
        hipLaunchKernelGGL((gc_exec_kernel<G, false, 4, kTpbWide>), dim3((L.nrec + per - 1) / per), dim3(per * 64), 0, st,
                           recs + L.first_rec, L.nrec, words, tab, dec, L.step0, R, w, p);
    } break;
    case LM_SPLIT:
        if constexpr (PART == 2)
            hipLaunchKernelGGL((gc_split_kernel<G>), dim3(L.nrec), dim3(1024), 0, st, recs + L.first_rec, L.nrec, words, tab, dec,
                               L.step0, R, w, p);
        break;
    case LM_QUAD2:
        if constexpr (PART == 3)
            hipLaunchKernelGGL((gc_exec_kernel<G, true, 2, 256>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec, L.nrec, words,
                               tab, dec, L.step0, R, w, p);
        break;
    }
    return hipGetLastError();
}

}  // namespace gc
