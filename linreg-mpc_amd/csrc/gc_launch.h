// gc_launch.h -- which kernel runs a launch of the garbled word machine, and with what geometry.
// One dispatch for the co-located solver (gc_engine.hip) and for the separate CSP / Evaluator
// objects (gc_roles.hip): garbler and evaluator of a launch must agree on the execution mode,
// because the mode fixes the order in which independent gate steps are numbered (B::kPairSteps).
#pragma once
#include <hip/hip_runtime.h>

#include "gc_device.h"
#include "gc_program.h"
#include "gc_split.h"

namespace gc {

// workgroup sizes of the MAC kernels (one workgroup per CU: 128 KiB of tables).  Measured on
// d=500: garbler 1024 threads (4 waves/SIMD, 128 VGPRs) and evaluator 768 threads are the fastest;
// 256-thread workgroups are 25 % slower
#ifndef GC_TPB_MACG
#define GC_TPB_MACG 1024
#endif
#ifndef GC_TPB_MACE
#define GC_TPB_MACE 768
#endif
static constexpr int kTpbMacG = GC_TPB_MACG, kTpbMacE = GC_TPB_MACE;
#ifndef GC_MAC_EXCLUSIVE
#define GC_MAC_EXCLUSIVE 1
#endif
// generic launches with at least this many records run one wave per record (throughput);
// narrower ones run one 4-wave workgroup per record (latency)
#ifndef GC_WIDE_LAUNCH
#define GC_WIDE_LAUNCH 2048
#endif
#ifndef GC_WIDE_ADAPT
#define GC_WIDE_ADAPT 1
#endif
static constexpr uint32_t kWideLaunch = GC_WIDE_LAUNCH;
// ... whatever the length of the records.  With an out-of-line gate body, long dependent records (dividers by the
// thousand in a merged lambda sweep) ran 5 % faster in the 4-wave mode and this was a bound on steps per record (256);
// with the gate bodies inlined (GC_SOLO_INLINE, gc_device.h) the one-wave-per-record kernel wins: 64-lambda sweep
// 8.39 -> 7.87 s.  The macro stays for A/B runs.
#ifndef GC_WIDE_MAX_STEPS
#define GC_WIDE_MAX_STEPS 100000000
#endif
static constexpr uint64_t kWideMaxSteps = GC_WIDE_MAX_STEPS;
// wide launches: 12 records (waves) per workgroup share one 128 KiB four-table image
#ifndef GC_TPB_WIDE
#define GC_TPB_WIDE 768
#endif
static constexpr int kTpbWide = GC_TPB_WIDE;
// 4-wave launches with at most one workgroup per CU use the four-table image (144 KiB per
// workgroup); larger ones the two-table 64 KiB image, so that two workgroups share a CU.
// The former are bound by the dependent chain of ONE record: their garbler runs the critical
// path only (2 hashes per gate) and a table pass completes the ciphertexts (gc_device.h: CRIT)
#ifndef GC_QUAD_ONE_PER_CU
#define GC_QUAD_ONE_PER_CU 256
#endif
static constexpr uint32_t kQuadOnePerCu = GC_QUAD_ONE_PER_CU;
#ifndef GC_CRIT
#define GC_CRIT 1
#endif
// ... and so do 4-wave launches of up to this many records (two workgroups per CU): their record kernel is
// still paced by the dependent chain, and the table pass runs beside the small launches that follow
#ifndef GC_CRIT_MAX_RECS
#define GC_CRIT_MAX_RECS 256   /* 1024 measured: no gain on d=500 CGD-15 nor on 8- and 64-circuit sweep blocks */
#endif
static constexpr uint32_t kCritMaxRecs = GC_CRIT_MAX_RECS;
// MAC launches with fewer records than this are latency-bound too (Cholesky / LDL^T stages at
// small d): they run in the 4-wave mode instead of the throughput-oriented MAC kernel
static constexpr uint32_t kNarrowMac = 1024;
// MAC launches of at least two garbler rounds get the chip to themselves (GC_MAC_EXCLUSIVE)
#ifndef GC_EXCLUSIVE_MAC_RECS
#define GC_EXCLUSIVE_MAC_RECS 8192
#endif
static constexpr uint32_t kExclusiveMac = GC_EXCLUSIVE_MAC_RECS;
static constexpr int kTpbTabfill = 1024;
// launches of at most one workgroup per CU run column-split on 16 waves per record (gc_split.h); the garbler side is
// critical-path garbling by construction, so it follows GC_CRIT
#ifndef GC_SPLIT
#define GC_SPLIT GC_CRIT
#endif
#ifndef GC_QUAD4
#define GC_QUAD4 0
#endif
// ... up to this many records (one workgroup per CU).  512, i.e. two rounds, measured on the 500-divider launches of
// d=500 CGD: 3.8 ms against 2.9 ms for the 4-wave kernel at two workgroups per CU -- with the whole chip busy the
// table pass of critical-path garbling is no longer free
#ifndef GC_SPLIT_MAX_RECS
#define GC_SPLIT_MAX_RECS 256
#endif
static constexpr uint32_t kSplitMaxRecs = GC_SPLIT_MAX_RECS;
// ... switchable per role at run time (lgc_set_split_kernels): the two kernels are interchangeable
inline int &gc_split_enabled(bool garbler) {
    static int on[2] = {1, 1};
    return on[garbler ? 0 : 1];
}

// MAC launches: one workgroup per CU and every record of a launch takes the same time, so a launch runs
// in rounds of (CUs x waves per workgroup) records and a partly filled last round costs a whole one.
// Pick the workgroup size (waves) in [lo, hi] that wastes the least: cost = rounds x waves.
#ifndef GC_MAC_TAIL_SPLIT
#define GC_MAC_TAIL_SPLIT 0   /* measured on d=100 (10 000 records): 0.067 -> 0.075 s of garbler MAC time: worse */
#endif
#ifndef GC_MAC_PERSIST_MAX_ROUNDS
#define GC_MAC_PERSIST_MAX_ROUNDS 8
#endif
#ifndef GC_MAC_ADAPT_LO_G
#define GC_MAC_ADAPT_LO_G 10
#endif
#ifndef GC_MAC_ADAPT_LO_E
#define GC_MAC_ADAPT_LO_E 8
#endif
#ifndef GC_MAC_ADAPT
#define GC_MAC_ADAPT 0   /* measured: -5 % on a serialised d=100 matvec, +10 % when it overlaps the evaluator chain */
#endif
static inline unsigned gc_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            cus = 256;
    }
    return (unsigned)cus;
}
static inline unsigned gc_mac_waves(uint32_t nrec, unsigned lo, unsigned hi) {
    if (!GC_MAC_ADAPT) return hi;
    const uint64_t cus = gc_num_cus();
    if (((uint64_t)nrec + hi - 1) / hi > 8 * cus) return hi;     // many rounds: the last one hardly matters, and the
                                                                 // cost model below (time of a round ~ waves) is only rough
    unsigned best = hi;
    uint64_t best_cost = ~0ull;
    for (unsigned wv = hi; wv >= lo; wv--) {
        uint64_t wgs = (nrec + wv - 1) / wv, rounds = (wgs + cus - 1) / cus, cost = rounds * wv;
        if (cost < best_cost) { best_cost = cost; best = wv; }
    }
    return best;
}

// the launch runs in the 16-wave column-split kernel (role G: garbler, else evaluator)
static inline bool gc_launch_is_split(const Launch &L, bool garbler) {
    const bool mac = L.mac_only && L.nrec >= kNarrowMac;
    const bool wide = L.nrec >= kWideLaunch && L.steps < (uint64_t)L.nrec * kWideMaxSteps;
    return GC_SPLIT && gc_split_enabled(garbler) && !mac && !wide && L.nrec > 0 && L.nrec <= kSplitMaxRecs;
}
// true when the garbler of this launch runs the critical path only and needs gc_launch_tabfill afterwards
static inline bool gc_launch_is_crit(const Launch &L) {
    const bool mac = L.mac_only && L.nrec >= kNarrowMac;
    const bool wide = L.nrec >= kWideLaunch && L.steps < (uint64_t)L.nrec * kWideMaxSteps;
    if (L.steps == 0) return false;
    if (gc_launch_is_split(L, true)) return true;
    return GC_CRIT && GC_QUAD4 && !mac && !wide && L.nrec > 0 && L.nrec <= kCritMaxRecs;
}
// the table pass of a critical-path launch; it only has to finish before the launch is EVALUATED, so the
// co-located solver runs it on a side stream while the garbler chain moves on to the next launch
static hipError_t gc_launch_tabfill(const Launch &L, Lbl *tab, Lbl R, hipStream_t st) {
    const uint64_t per = kTpbTabfill / 64;
    uint64_t blocks = (L.steps + per - 1) / per;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((gc_tabfill_kernel<kTpbTabfill>), dim3((unsigned)blocks), dim3(kTpbTabfill), 0, st, tab,
                       (uint32_t)L.steps, L.step0, R);
    return hipGetLastError();
}

// the record kernel of a launch (garbler: without the table pass)
template <bool G>
static hipError_t gc_launch_records(const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w, int p,
                                    hipStream_t st) {
    if (L.nrec == 0) return hipSuccess;
    if (L.mac_only && L.nrec >= kNarrowMac) {
        constexpr int TPB = G ? kTpbMacG : kTpbMacE;       // upper bound (register budget of the kernel)
        const unsigned per = gc_mac_waves(L.nrec, G ? GC_MAC_ADAPT_LO_G : GC_MAC_ADAPT_LO_E, TPB / 64);
#if GC_MAC_TAIL_SPLIT
        // One workgroup per CU, every record the same length: the launch runs in rounds of (CUs x waves) records and
        // a partly filled last round costs a whole one.  The records beyond the last full round therefore go into a
        // second launch of ONE workgroup per CU with just enough waves: an LDS-bound workgroup of w waves takes
        // about w / 16 of the time of a full one, so the tail costs its share instead of a round.
        const uint32_t round = gc_num_cus() * per, full = L.nrec / round * round, rest = L.nrec - full;
        if (!GC_MAC_ADAPT && full && rest && rest < round - round / 8) {
            unsigned wv = (rest + gc_num_cus() - 1) / gc_num_cus();
            hipLaunchKernelGGL((gc_mac_kernel<G, TPB>), dim3(full / per), dim3(per * 64), 0, st, recs + L.first_rec, full, words,
                               tab, L.step0, R, w, p);
            hipLaunchKernelGGL((gc_mac_kernel<G, TPB>), dim3((rest + wv - 1) / wv), dim3(wv * 64), 0, st, recs + L.first_rec + full, rest,
                               words, tab, L.step0, R, w, p);
            return hipGetLastError();
        }
#endif
        unsigned wgs = (L.nrec + per - 1) / per;
#if GC_MAC_PERSIST      /* one workgroup per CU: the waves walk the records themselves (gc_device.h) */
        if (G && wgs > gc_num_cus() && wgs <= GC_MAC_PERSIST_MAX_ROUNDS * gc_num_cus()) wgs = gc_num_cus();
#endif
        hipLaunchKernelGGL((gc_mac_kernel<G, TPB>), dim3(wgs), dim3(per * 64), 0, st, recs + L.first_rec, L.nrec, words, tab,
                           L.step0, R, w, p);
    } else if (L.nrec >= kWideLaunch && L.steps < (uint64_t)L.nrec * kWideMaxSteps) {
        // records (waves) per workgroup: as few as keep the launch within one workgroup per CU, at most TPB / 64 --
        // a launch of 800 dividers runs as 200 workgroups of 4 waves, one round, instead of 67 CUs with 12 waves each
        unsigned per = (L.nrec + gc_num_cus() - 1) / gc_num_cus();
        if (per > (unsigned)kTpbWide / 64) per = kTpbWide / 64;
        if (!GC_WIDE_ADAPT) per = kTpbWide / 64;
        hipLaunchKernelGGL((gc_exec_kernel<G, false, 4, kTpbWide>), dim3((L.nrec + per - 1) / per), dim3(per * 64), 0, st,
                           recs + L.first_rec, L.nrec, words, tab, dec, L.step0, R, w, p);
    } else if (gc_launch_is_split(L, G)) {
        hipLaunchKernelGGL((gc_split_kernel<G>), dim3(L.nrec), dim3(1024), 0, st, recs + L.first_rec, L.nrec, words, tab, dec, L.step0,
                           R, w, p);
#if GC_QUAD4          /* the 4-wave kernels with the four-table image (garbler: critical-path garbling): what ran these launches
                         before the column-split kernel; with 0 (default, 45 s less to compile) a role whose split kernel is
                         switched off runs them in the two-table 4-wave kernel below */
    } else if (L.nrec <= kQuadOnePerCu) {
        hipLaunchKernelGGL((gc_exec_kernel<G, true, 4, 256, G && GC_CRIT>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec,
                           L.nrec, words, tab, dec, L.step0, R, w, p);
#endif
#if GC_CRIT_MAX_RECS > GC_QUAD_ONE_PER_CU     /* critical-path garbling at two workgroups per CU: measured, no gain; not instantiated */
    } else if (L.nrec <= kCritMaxRecs) {
        hipLaunchKernelGGL((gc_exec_kernel<G, true, 2, 256, G && GC_CRIT>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec,
                           L.nrec, words, tab, dec, L.step0, R, w, p);
#endif
    } else {
        hipLaunchKernelGGL((gc_exec_kernel<G, true, 2, 256>), dim3(L.nrec), dim3(256), 0, st, recs + L.first_rec, L.nrec, words,
                           tab, dec, L.step0, R, w, p);
    }
    return hipGetLastError();
}

template <bool G>
static hipError_t gc_launch(const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w, int p,
                            hipStream_t st) {
    hipError_t e = gc_launch_records<G>(recs, L, words, dec, tab, R, w, p, st);
    if (e == hipSuccess && G && gc_launch_is_crit(L)) e = gc_launch_tabfill(L, tab, R, st);
    return e;
}

}  // namespace gc
