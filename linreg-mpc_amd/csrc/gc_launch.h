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
#ifndef GC_TPB_MACKG
#define GC_TPB_MACKG 1024
#endif
#ifndef GC_TPB_MACKE
#define GC_TPB_MACKE 768
#endif
static constexpr int kTpbMackG = GC_TPB_MACKG, kTpbMackE = GC_TPB_MACKE;
// ... and over the table-free gate hash (lgc_set_gate_hash(1)): no 128 KiB table image ties a workgroup to a CU, the
// register budget alone sets the occupancy
#ifndef GC_TPB_MACG_H1
#define GC_TPB_MACG_H1 1024
#endif
#ifndef GC_TPB_MACE_H1
#define GC_TPB_MACE_H1 768
#endif
#ifndef GC_TPB_MACKG_H1
#define GC_TPB_MACKG_H1 512    /* scripts/exp/hash_ab.sh: 512 / 512 d=500 CGD-15 0.923 s, 1024 / 768 0.940 s, 256 / 256 0.921 s */
#endif
#ifndef GC_TPB_MACKE_H1
#define GC_TPB_MACKE_H1 512
#endif
// Unused dynamic LDS per workgroup of the table-free MACK kernels: caps their occupancy so that the other role's small
// launches find registers on every CU.  Measured (scripts/exp/hash_ab.sh, d=500 CGD-15 over gate hash 1): 96 KiB with 512
// threads (8 waves per CU) MACK +7 %, solve 0.938 s against 0.927-0.932 s -- what the small launches gain the MACK loses;
// 4 waves per CU: MACK +12 %.  0 = off.
#ifndef GC_MACK_PAD_H1
#define GC_MACK_PAD_H1 0
#endif
static constexpr unsigned kMackPadH1 = GC_MACK_PAD_H1;
static constexpr int kTpbMacGH = GC_TPB_MACG_H1, kTpbMacEH = GC_TPB_MACE_H1, kTpbMackGH = GC_TPB_MACKG_H1, kTpbMackEH = GC_TPB_MACKE_H1;
#ifndef GC_MAC_EXCLUSIVE
#define GC_MAC_EXCLUSIVE 1
#endif
// generic launches with at least this many records run one wave per record (throughput);
// narrower ones run one 4-wave workgroup per record (latency)
#ifndef GC_WIDE_LAUNCH
#define GC_WIDE_LAUNCH 520   /* beyond the 512 records that two 4-wave workgroups per CU hold at once (the 800 dividers of an
                                8-lambda block would take two rounds there).  Round 2 had 2048: with Kogge-Stone adders a level
                                was a dual step, which the 4-wave kernel splits over its waves; the levels of the Sklansky
                                adder are single steps.  scripts/dbg/wide_ab.sh: 8-lambda block 0.861 -> 0.826 s */
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
#define GC_EXCLUSIVE_MAC_RECS 4096   /* one full garbler round (the equal pieces of a merged sweep launch may be ~8000 records) */
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
// ... switchable per role at run time (lgc_set_split_kernels): the two kernels are interchangeable.  A launch reads the
// flag ONCE (gc_launch_mode) and passes the decision down: record kernel and table pass can never disagree
inline std::atomic<int> &gc_split_enabled(bool garbler) {
    static std::atomic<int> on[2] = {{1}, {1}};
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
// Waves per workgroup of a plain MAC launch of a few rounds (gc_mac_waves).  Round 2 let it go down to 10 (garbler) / 8
// (evaluator) waves: -5 % on a serialised d = 100 matrix-vector product, +10 % beside the evaluator chain, and it was switched
// off.  With a floor of 12 waves it pays (end of round 5, scripts/exp/mac_adapt12_ab.sh, A B A B on one box): the 10 000
// records of a d = 100 product run as three rounds of 14 waves instead of 16 + 16 + 7 -- d = 100 CGD-15 0.1335 -> 0.1277 s,
// d = 40 0.062 -> 0.058, 32-bit d = 100 0.0575 -> 0.0528, Cholesky d = 100 0.328 -> 0.324; d = 64 / 120 / 180 / 200 / 250 and
// the 32-bit d = 300 within +-1 %.  (The floor of 10 picks 10 waves x 4 rounds at d = 100 and is 6 % SLOWER: a round's time
// stops falling with its waves below a dozen, and every round pays a table fill.)  Launches of more than eight rounds are
// left alone; the Karatsuba launches have their own switch (GC_MACK_ADAPT, off: measured slower).
#ifndef GC_MAC_ADAPT_LO_G
#define GC_MAC_ADAPT_LO_G 12
#endif
#ifndef GC_MAC_ADAPT_LO_E
#define GC_MAC_ADAPT_LO_E 12
#endif
#ifndef GC_MAC_ADAPT
#define GC_MAC_ADAPT 1
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
// Round 5 tried this on launches of up to ten rounds with 12 (garbler) / 9 (evaluator) waves at least
// (scripts/exp/mack_adapt_ab.sh): the serialised MAC time of a d = 500 factorisation fell and the overlapped run got slower
// (11.7 -> 12.2 s) -- a column's MAC launch runs beside the other role's chain, and a partly filled round is CUs for that
// chain.  On for launches of at most THREE rounds and the garbler only (the evaluator's workgroups are twelve waves already):
// what is left are the matrix-vector products of small CGD systems, which their chain waits for (scripts/exp/kara_small_ab.sh:
// d = 100 CGD-15 0.128 -> 0.122 s together with Karatsuba records at that size, d = 250 Cholesky 1.97 -> 1.96 s).
#ifndef GC_MACK_ADAPT
#define GC_MACK_ADAPT 1
#endif
#ifndef GC_MACK_ADAPT_LO_G
#define GC_MACK_ADAPT_LO_G 12
#endif
#ifndef GC_MACK_ADAPT_LO_E
#define GC_MACK_ADAPT_LO_E 12
#endif
#ifndef GC_MACK_ADAPT_MAX_ROUNDS
#define GC_MACK_ADAPT_MAX_ROUNDS 3
#endif
// waves per workgroup of a Karatsuba MAC launch of a few rounds: the count with the least rounds x waves
static inline unsigned gc_mack_waves(uint32_t nrec, unsigned lo, unsigned hi) {
    const uint64_t cus = gc_num_cus();
    if (((uint64_t)nrec + hi - 1) / hi > (uint64_t)GC_MACK_ADAPT_MAX_ROUNDS * cus || nrec <= cus * lo) return hi;
    unsigned best = hi;
    uint64_t best_cost = ~0ull;
    for (unsigned wv = hi; wv >= lo; wv--) {
        uint64_t wgs = (nrec + wv - 1) / wv, rounds = (wgs + cus - 1) / cus, cost = rounds * wv;
        if (cost < best_cost) { best_cost = cost; best = wv; }
    }
    return best;
}
// ... and of a wide launch (one wave per generic record) of a few rounds.  Records of one launch are mostly of one length and
// the workgroups are bound by the LDS lookups of their waves together, so a round of w-wave workgroups takes about w / hi of a
// full one: 6 400 dividers (64 circuits x d = 100) are 3 rounds of 12 waves -- the third almost empty -- or 3 rounds of 9.
// GC_WIDE_ADAPT_LO: the fewest waves tried (LGC_WIDE_LO in the environment overrides it for A/B runs; 12 = off).
#ifndef GC_WIDE_ADAPT_LO
#define GC_WIDE_ADAPT_LO 12
#endif
static inline unsigned gc_wide_waves(uint32_t nrec, unsigned hi) {
    static const unsigned lo_env = [] { const char *e = getenv("LGC_WIDE_LO"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : (unsigned)GC_WIDE_ADAPT_LO; }();
    const unsigned lo = lo_env < hi ? lo_env : hi;
    const uint64_t cus = gc_num_cus();
    if (((uint64_t)nrec + hi - 1) / hi > 12 * cus) return hi;
    unsigned best = hi;
    uint64_t best_cost = ~0ull;
    for (unsigned wv = hi; wv >= lo; wv--) {
        uint64_t wgs = (nrec + wv - 1) / wv, rounds = (wgs + cus - 1) / cus, cost = rounds * wv;
        if (cost < best_cost) { best_cost = cost; best = wv; }
    }
    return best;
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

// which kernel runs a launch for one role
enum LaunchMode {
    LM_NONE = 0,     // no records
    LM_MAC,          // gc_mac_kernel: one wave per record, throughput
    LM_MACK,         // gc_mack_kernel: the same for OP_MACK records (Karatsuba products)
    LM_WIDE,         // gc_exec_kernel<.., false, 4, ..>: one wave per generic record
    LM_SPLIT,        // gc_split_kernel: 16 waves per record, column-split (garbler: critical path + table pass)
    LM_QUAD4,        // GC_QUAD4 builds only: 4 waves per record on the four-table image (garbler: critical path + table pass)
    LM_QUAD2         // 4 waves per record on the two-table image, two workgroups per CU
};
// hash: the program's gate hash (Program::gate_hash).  The column-split and the critical-path kernels hash in AES layouts
// of their own: with the table-free hash (1) the narrow launches run in the plain 4-wave kernel
static inline LaunchMode gc_launch_mode(const Launch &L, bool garbler, int hash) {
    if (L.nrec == 0) return LM_NONE;
    if (L.mac_only && L.nrec >= kNarrowMac) return L.mack ? LM_MACK : LM_MAC;
    if (L.nrec >= kWideLaunch && L.steps < (uint64_t)L.nrec * kWideMaxSteps) return LM_WIDE;
    if (hash != GATE_HASH_AES) return LM_QUAD2;
    if (GC_SPLIT && L.nrec <= kSplitMaxRecs && gc_split_enabled(garbler).load(std::memory_order_relaxed)) return LM_SPLIT;
    if (GC_QUAD4 && L.nrec <= kQuadOnePerCu) return LM_QUAD4;
    return LM_QUAD2;
}
// true when the garbler's record kernel of this mode computes the critical path only: it leaves the zero-labels
// (a0, b0) of every gate in a STASH (two rows per gate step, the layout of the launch's table) and gc_launch_tabfill
// turns the stash into the ciphertexts
static inline bool gc_mode_is_crit(LaunchMode m, const Launch &L) {
    return L.steps != 0 && (m == LM_SPLIT || (m == LM_QUAD4 && GC_CRIT));
}
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
GC_KERN_DECL(g_0h) GC_KERN_DECL(g_1h) GC_KERN_DECL(g_3h) GC_KERN_DECL(e_0h) GC_KERN_DECL(e_1h) GC_KERN_DECL(e_3h)      // gate hash 1
#undef GC_KERN_DECL
// the record kernel of a launch in mode `m` (garbler in a critical-path mode: `tab` is the stash)
template <bool G>
static inline hipError_t gc_launch_records(LaunchMode m, int hash, const Rec *recs, const Launch &L, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R,
                                          int w, int p, hipStream_t st) {
    if (hash == GATE_HASH_CHASKEY12) {
        switch (m) {
        case LM_NONE: return hipSuccess;
        case LM_MAC: case LM_MACK: return G ? gc_launch_records_g_0h(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_0h(m, recs, L, words, dec, tab, R, w, p, st);
        case LM_WIDE: return G ? gc_launch_records_g_1h(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_1h(m, recs, L, words, dec, tab, R, w, p, st);
        case LM_QUAD2: return G ? gc_launch_records_g_3h(m, recs, L, words, dec, tab, R, w, p, st) : gc_launch_records_e_3h(m, recs, L, words, dec, tab, R, w, p, st);
        default: return hipErrorInvalidValue;      // gc_launch_mode never picks an AES-only kernel for this hash
        }
    }
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
static hipError_t gc_preload(const std::vector<Launch> &launches, int hash, hipStream_t st) {
    bool need[4] = {false, false, false, false};
    for (const Launch &L : launches) {
        switch (gc_launch_mode(L, G, hash)) {
        case LM_NONE: break;
        case LM_MAC: case LM_MACK: need[0] = true; break;
        case LM_WIDE: need[1] = true; break;
        case LM_SPLIT: need[2] = true; break;
        default: need[3] = true; break;
        }
    }
    hipError_t e = hipSuccess;
    if (hash == GATE_HASH_CHASKEY12) {
        if (need[0] && e == hipSuccess) e = G ? gc_kern_touch_g_0h(st) : gc_kern_touch_e_0h(st);
        if (need[1] && e == hipSuccess) e = G ? gc_kern_touch_g_1h(st) : gc_kern_touch_e_1h(st);
        if ((need[2] || need[3]) && e == hipSuccess) e = G ? gc_kern_touch_g_3h(st) : gc_kern_touch_e_3h(st);
        return e;
    }
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
static hipError_t gc_launch(const Rec *recs, const Launch &L, int hash, Lbl *words, uint64_t *dec, Lbl *tab, Lbl R, int w, int p,
                            hipStream_t st, Lbl *stash = 0, int stages = 3, bool *was_crit = 0) {
    const LaunchMode m = gc_launch_mode(L, G, hash);
    const bool crit = G && gc_mode_is_crit(m, L);
    if (was_crit) *was_crit = crit;
    Lbl *rec_tab = (crit && stash) ? stash : tab;
    hipError_t e = hipSuccess;
    if (stages & 1) e = gc_launch_records<G>(m, hash, recs, L, words, dec, rec_tab, R, w, p, st);
    if (e == hipSuccess && crit && (stages & 2)) e = gc_launch_tabfill(L, rec_tab, tab, R, st);
    return e;
}

}  // namespace gc
