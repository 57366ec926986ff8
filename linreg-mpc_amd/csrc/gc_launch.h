// gc_launch.h -- which kernel runs a launch of the garbled word machine, and with what geometry.
// One dispatch for the co-located solver (gc_engine.hip) and for the separate CSP / Evaluator
// objects (gc_roles.hip): garbler and evaluator of a launch must agree on the execution mode,
// because the mode fixes the order in which independent gate steps are numbered (B::kPairSteps).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <vector>

#include "gc_aes.h"
#include "gc_program.h"

namespace gc {

// Workgroup sizes of the MAC kernels (one workgroup per CU: 128 KiB of tables).  Measured on d = 500: garbler 1024 threads
// (4 waves/SIMD, 128 VGPRs) and evaluator 768 threads are the fastest; 256-thread workgroups are 25 % slower.  The four
// sizes and the two launch-width thresholds below are the tunables a -D may override; everything else is a constant.
#ifndef GC_TPB_MACG
#define GC_TPB_MACG 1024
#endif
#ifndef GC_TPB_MACE
#define GC_TPB_MACE 768
#endif
#ifndef GC_TPB_MACKG
#define GC_TPB_MACKG 1024
#endif
#ifndef GC_TPB_MACKE
#define GC_TPB_MACKE 768
#endif
static constexpr int kTpbMacG = GC_TPB_MACG, kTpbMacE = GC_TPB_MACE, kTpbMackG = GC_TPB_MACKG, kTpbMackE = GC_TPB_MACKE;
// generic launches with at least this many records run one wave per record (throughput); narrower ones run one multi-wave
// workgroup per record (latency).  520: beyond the 512 records that two 4-wave workgroups per CU hold at once (the 800
// dividers of an 8-lambda block would take two rounds there).
#ifndef GC_WIDE_LAUNCH
#define GC_WIDE_LAUNCH 520
#endif
static constexpr uint32_t kWideLaunch = GC_WIDE_LAUNCH;
// wide launches: 12 records (waves) per workgroup share one 128 KiB four-table image
static constexpr int kTpbWide = 768;
// MAC launches with fewer records than this are latency-bound too (Cholesky / LDL^T stages at small d): they run in the
// multi-wave modes instead of the throughput-oriented MAC kernel
static constexpr uint32_t kNarrowMac = 1024;
// MAC launches of at least one full garbler round get the chip to themselves: garble k, evaluate k, garble k + 1 (the equal
// pieces of a merged sweep launch may be ~8000 records)
static constexpr uint32_t kExclusiveMac = 4096;
static constexpr int kTpbTabfill = 1024;
// Launches of at most this many records (one workgroup per CU) run column-split on 16 waves per record (gc_split.h); the
// garbler side is critical-path garbling: output labels only, the ciphertexts by gc_tabfill_kernel.  512, i.e. two rounds,
// measured on the 500-divider launches of d = 500 CGD: 3.8 ms against 2.9 ms for the 4-wave kernel at two workgroups per CU
// -- with the whole chip busy the table pass is no longer free.
#ifndef GC_SPLIT_MAX_RECS
#define GC_SPLIT_MAX_RECS 256
#endif
static constexpr uint32_t kSplitMaxRecs = GC_SPLIT_MAX_RECS;
// ... switchable per role at run time (lgc_set_split_kernels, linreg_gc_debug.h: the interchangeability test): the two
// kernels are interchangeable.  A launch reads the flag ONCE (gc_launch_mode) and passes the decision down: record kernel and
// table pass can never disagree
inline std::atomic<int> &gc_split_enabled(bool garbler) {
    static std::atomic<int> on[2] = {{1}, {1}};
    return on[garbler ? 0 : 1];
}

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
// MAC launches: one workgroup per CU and every record of a launch takes the same time, so a launch runs in rounds of
// (CUs x waves per workgroup) records and a partly filled last round costs a whole one.  A round's time falls with its waves
// down to about a dozen: pick the workgroup size (waves) in [kMacAdaptLo, hi] with the least rounds x waves.  The 10 000
// records of a d = 100 product run as three rounds of 14 waves instead of 16 + 16 + 7 (d = 100 CGD-15 0.1335 -> 0.1277 s;
// a floor of 10 picks 10 waves x 4 rounds and is 6 % slower: every round pays a table fill).  Plain MAC launches of more
// than eight rounds are left alone; Karatsuba launches of more than three (a column's MAC launch runs beside the other role's
// chain, and a partly filled round is CUs for that chain: d = 500 factorisations 11.7 -> 12.2 s with the limit at ten).
static constexpr unsigned kMacAdaptLo = 12, kMacAdaptMaxRounds = 8, kMackAdaptMaxRounds = 3;
static inline unsigned gc_adapt_waves(uint32_t nrec, unsigned lo, unsigned hi, unsigned max_rounds) {
    const uint64_t cus = gc_num_cus();
    if (((uint64_t)nrec + hi - 1) / hi > (uint64_t)max_rounds * cus || nrec <= cus * lo) return hi;
    unsigned best = hi;
    uint64_t best_cost = ~0ull;
    for (unsigned wv = hi; wv >= lo; wv--) {
        uint64_t wgs = (nrec + wv - 1) / wv, rounds = (wgs + cus - 1) / cus, cost = rounds * wv;
        if (cost < best_cost) { best_cost = cost; best = wv; }
    }
    return best;
}
// Records per WORKGROUP of a MAC launch (MacQueue, gc_device.h).  A SIMD
// issues from its OLDEST wave first: of sixteen waves with one record each, the four that arrived first on their SIMDs are
// through after 45 % of the workgroup's time and the CU hashes with twelve, eight, four waves until the last four are done
// (profiles/r6_wave_order.txt).  So the waves of a workgroup PULL records from a chunk the workgroup owns, and every wave
// stays busy to the end whatever its speed.  How large a chunk: a launch of equal workgroups runs in rounds (one workgroup
// per CU) and ends with a partly filled one -- the longer the workgroups the more that costs -- while the OTHER role's small
// launches get CUs only when workgroups retire.  Measured on d = 500 CGD-15 (scripts/gpu_headline.py, one box, s per solve):
// 1 record per wave 1.769; 4: 1.731; 8: 1.694; 12: 1.664; 16 (garbler 2 rounds, evaluator 3): 1.659; 24-200 (ONE round: the
// launch is persistent) 1.645-1.653 -- an iteration is then garble + evaluate + the two chains together on an idle chip
// (65.2 + 37.0 + 7.5 ms) instead of 74.3 + 39.1 + 4.5; another box: 1.791 -> 1.741 (2 / 3 rounds) -> 1.726 (one round);
// d = 300: 0.691 -> 0.653-0.667; 32-bit d = 500 CGD-20: 0.762 -> 0.740.  So: about kMacChunk records per wave, the number of
// workgroups a whole number of rounds, at least one; launches of less than one round of one record per wave keep the static
// assignment.  (Launches of one to four rounds, the products of mid-sized systems: d = 100 CGD-15 0.119 -> 0.1155 s, d = 150
// 0.215 -> 0.205, d = 200 and the 32-bit d = 300 within 1 %, Cholesky d = 250 1.956 -> 1.920.)
static constexpr unsigned kMacChunk = 32, kMacChunkMinRounds = 1;
static inline unsigned gc_mac_per_wg(uint32_t nrec, unsigned waves) {
    const uint64_t cus = gc_num_cus();
    if ((uint64_t)nrec < (uint64_t)kMacChunkMinRounds * cus * waves) return waves;
    uint64_t rounds = ((uint64_t)nrec + cus * waves * kMacChunk / 2) / (cus * waves * kMacChunk);      // nearest whole number of rounds
    if (rounds < 1) rounds = 1;
    const uint64_t wgs = rounds * cus;
    return (unsigned)(((uint64_t)nrec + wgs - 1) / wgs);
}
static inline unsigned gc_mac_waves(uint32_t nrec, unsigned lo, unsigned hi) { return gc_adapt_waves(nrec, lo, hi, kMacAdaptMaxRounds); }
static inline unsigned gc_mack_waves(uint32_t nrec, unsigned lo, unsigned hi) { return gc_adapt_waves(nrec, lo, hi, kMackAdaptMaxRounds); }

// which kernel runs a launch for one role
enum LaunchMode {
    LM_NONE = 0,     // no records
    LM_MAC,          // gc_mac_kernel: one wave per record, throughput
    LM_MACK,         // gc_mack_kernel: the same for OP_MACK records (Karatsuba products)
    LM_WIDE,         // gc_exec_kernel<.., false, 4, ..>: one wave per generic record
    LM_SPLIT,        // gc_split_kernel: 16 waves per record, column-split (garbler: critical path + table pass)
    LM_QUAD2         // 4 waves per record on the two-table image, two workgroups per CU
};
static inline LaunchMode gc_launch_mode(const Launch &L, bool garbler) {
    if (L.nrec == 0) return LM_NONE;
    if (L.mac_only && L.nrec >= kNarrowMac) return L.mack ? LM_MACK : LM_MAC;
    if (L.nrec >= kWideLaunch) return LM_WIDE;
    if (L.nrec <= kSplitMaxRecs && gc_split_enabled(garbler).load(std::memory_order_relaxed)) return LM_SPLIT;
    return LM_QUAD2;
}
// true when the garbler's record kernel of this mode computes the critical path only: it leaves the zero-labels
// (a0, b0) of every gate in a STASH (two rows per gate step, the layout of the launch's table) and gc_launch_tabfill
// turns the stash into the ciphertexts
static inline bool gc_mode_is_crit(LaunchMode m, const Launch &L) { return L.steps != 0 && m == LM_SPLIT; }
// The kernels live in translation units of their own -- gc_kern.hip compiled per (role, kernel family), gc_kernels.h
// holds the dispatch -- so that the library builds in parallel.
// The table pass of a critical-path launch: stash -> tab (in place when they are the same buffer).  It only has to
// finish before the launch is EVALUATED, so the co-located solver runs it on a side stream while the garbler chain
// moves on to the next launch.  With the roles in different processes `tab` is memory the evaluator maps (hipIpc
// table ring) and `stash` is private to the garbler: nothing but finished ciphertexts is ever stored to `tab`.
hipError_t gc_launch_tabfill(const Launch &L, const Lbl *stash, Lbl *tab, Lbl R, hipStream_t st);
#define GC_KERN_DECL(tag)                                                                                                              \
    hipError_t gc_launch_records_##tag(LaunchMode m, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w, \
                                       int p, hipStream_t st);                                                                         \
    hipError_t gc_kern_touch_##tag(hipStream_t st);
GC_KERN_DECL(g_0) GC_KERN_DECL(g_1) GC_KERN_DECL(g_2) GC_KERN_DECL(g_3) GC_KERN_DECL(e_0) GC_KERN_DECL(e_1) GC_KERN_DECL(e_2) GC_KERN_DECL(e_3)
#undef GC_KERN_DECL
// the record kernel of a launch in mode `m` (garbler in a critical-path mode: `tab` is the stash)
template <bool G>
static inline hipError_t gc_launch_records(LaunchMode m, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R,
                                          int w, int p, hipStream_t st) {
    switch (m) {
    case LM_NONE: return hipSuccess;
    case LM_MAC: case LM_MACK: return G ? gc_launch_records_g_0(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_0(m, recs, L, words, dec, tab, R, w, p, st);
    case LM_SPLIT: return G ? gc_launch_records_g_2(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_2(m, recs, L, words, dec, tab, R, w, p, st);
    case LM_WIDE: return G ? gc_launch_records_g_1(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_1(m, recs, L, words, dec, tab, R, w, p, st);
    default: return G ? gc_launch_records_g_3(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_3(m, recs, L, words, dec, tab, R, w, p, st);
    }
}

// Loads the code objects a program's launches will run from, for one role, ahead of the first launch (one empty kernel per
// translation unit; a code object is otherwise loaded inside the first launch that needs it -- 5-10 ms each, on the critical
// path of a short run)
template <bool G>
static hipError_t gc_preload(const std::vector<Launch> &launches, hipStream_t st) {
    bool need[4] = {false, false, false, false};
    for (const Launch &L : launches) {
        switch (gc_launch_mode(L, G)) {
        case LM_NONE: break;
        case LM_MAC: case LM_MACK: need[0] = true; break;
        case LM_WIDE: need[1] = true; break;
        case LM_SPLIT: need[2] = true; break;
        default: need[3] = true; break;
        }
    }
    hipError_t e = hipSuccess;
    if (need[0] && e == hipSuccess) e = G ? gc_kern_touch_g_0(st) : gc_kern_touch_e_0(st);
    if (need[1] && e == hipSuccess) e = G ? gc_kern_touch_g_1(st) : gc_kern_touch_e_1(st);
    if (need[2] && e == hipSuccess) e = G ? gc_kern_touch_g_2(st) : gc_kern_touch_e_2(st);
    if (need[3] && e == hipSuccess) e = G ? gc_kern_touch_g_3(st) : gc_kern_touch_e_3(st);
    return e;
}

// A whole launch of one role.  stash: garbler only -- where a critical-path record kernel leaves the zero-labels for the
// table pass; 0 = in the launch's own table rows (the co-located solver, whose ring no other process maps).
// stages (test hook only): 1 = record kernel, 2 = table pass, 3 = both; *was_crit reports whether the launch has a table pass.
template <bool G>
static hipError_t gc_launch(const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w, int p,
                            hipStream_t st, Lbl *stash = 0, int stages = 3, bool *was_crit = 0) {
    const LaunchMode m = gc_launch_mode(L, G);
    const bool crit = G && gc_mode_is_crit(m, L);
    if (was_crit) *was_crit = crit;
    Lbl *rec_tab = (crit && stash) ? stash : tab;
    hipError_t e = hipSuccess;
    if (stages & 1) e = gc_launch_records<G>(m, recs, L, words, dec, rec_tab, R, w, p, st);
    if (e == hipSuccess && crit && (stages & 2)) e = gc_launch_tabfill(L, rec_tab, tab, R, st);
    return e;
}

}  // namespace gc
