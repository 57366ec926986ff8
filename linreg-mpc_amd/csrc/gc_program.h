// gc_program.h -- host-side lowering of the phase-2 solvers to launches of
// word-machine records (no device code here).
//
// Follows, statement by statement:
//   input assembly  src/linear.oc:10-94 (data providers) / :96-135 (two-party)
//   cgd             src/cgd.oc:96-212
//   cholesky        src/cholesky.oc:51-93
//   ldlt            src/ldlt.oc:50-90
// Wrap-around additions are re-associated freely (tree / carry-save sums):
// addition mod 2^w is associative, so results are identical.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>

#include "gc_exec.h"

namespace gc {

static const size_t kMinRecsPerLaunch = 8192;   // >= 2x the chip's resident waves (256 CUs x 12-16)
// Largest launch of the co-located solver, in gate steps (2 KiB of garbled table each): 2^25 = 64 GiB of tables.  A MAC
// launch runs in whole rounds of the chip (one workgroup per CU: 4096 garbler / 3072 evaluator records per round), so the
// fewer launches a matrix-vector product is cut into, the less is lost to partly filled last rounds: with 2^23 the d = 500
// product (27.5 M steps with Karatsuba records) fell into four launches of 3.81 / 5.09 rounds -- the evaluator ran six
// rounds for five rounds of work -- with 2^25 it is ONE launch.  (Round 4: with records of one Karatsuba pair a launch is tens
// of rounds and the partly filled last round no longer matters, but every launch still pays ~1 ms of ramp-up and tail, and
// smaller launches measured slower: d = 500 CGD-15 1.81 s at 2^25, 1.87 at 2^24, 1.93 at 2^23; scripts/exp/mac_ab.sh cap*.)
// The table ring is the largest launch plus kRingSlackBytes: 64 of the 288 GB of an MI355X at d = 500.
#ifndef GC_DEFAULT_CAP_LOG2
#define GC_DEFAULT_CAP_LOG2 25
#endif
static const uint64_t kDefaultCapSteps = 1ull << GC_DEFAULT_CAP_LOG2;
// ... a merged sweep program is cut into equal pieces by the same cap (replicate_program).  Rounds 3 - 4 used 2^24 there, so
// that the ring of a sweep block (32 + 8 GiB) fitted the one a d = 500 solver leaves parked; but every launch pays ~1 ms of
// ramp-up and tail (all workgroups start in lock step, the last ones leave CUs idle), and a 64-circuit block has 90 such
// launches: 2^25 halves them -- 6.27 -> 6.00 s for the 64-lambda block (scripts/exp/sweep_cap_ab.sh); blocks of <= 16 circuits
// fit one launch per product either way.
#ifndef GC_SWEEP_CAP_LOG2
#define GC_SWEEP_CAP_LOG2 25
#endif
static const uint64_t kSweepCapSteps = 1ull << GC_SWEEP_CAP_LOG2;
static const uint64_t kGenericCapSteps = 1ull << 24;     // launches other than MAC launches (Program::emit)
static const size_t kRingSlackBytes = (size_t)8 << 30;
// ... adjustable (lgc_set_table_ring_slack): eight ranks rehearsing an 8-GPU sweep on ONE GPU must fit its HBM together
inline size_t &ring_slack_bytes() { static size_t v = kRingSlackBytes; return v; }

struct Launch {
    uint32_t first_rec, nrec;   // slice of Program::recs
    uint64_t step0, steps;      // gate steps covered
    uint64_t gates;             // active AND gates
    bool mac_only;
    bool mack;                  // mac_only launch whose records are OP_MACK (Karatsuba products: a kernel of their own)
};

enum Alg { ALG_CHOLESKY = 0, ALG_LDLT = 1, ALG_CGD = 2, ALG_DIMCHECK = 3 };

struct Program {
    int w, p;
    size_t d, T, nshares;
    std::vector<Rec> recs;
    std::vector<Launch> launches;
    uint32_t n_words;            // word file size
    uint32_t n_reveal;           // decode slots
    uint32_t in_base;            // first input word: nshares x (T + d), share-major
    uint32_t rv_beta;            // decode slot of beta[0]
    uint32_t rv_trace;           // decode slot of trace[0] (cgd: iters x (d+4)), or ~0u
    uint32_t rv_ab;              // decode slot of the debug reveal of a, b (T + d), or ~0u
    uint64_t total_steps, total_gates;
    uint64_t total_xors = 0;     // flat-list XOR gates (rec_cost): reporting only
    uint64_t max_launch_steps;
    // cgd: per iteration, the last launch of the iteration and the AND gates emitted up to there
    // (the points where cgd.oc:190-194 prints yaoGateCount() and the running time)
    std::vector<uint32_t> iter_launch;
    std::vector<uint64_t> iter_gates;
    // lambda sweep (replicate_program): `replicas` copies of one circuit in one program; copy t
    // uses words x + t * word_stride (x != 0) and decode slots r + t * reveal_stride
    uint32_t replicas, word_stride, reveal_stride;
    uint32_t lam_rec;            // index of the OP_CONST record holding lambda, or ~0u
    // shared prefix of a sweep (data-provider path): words [0, shared_end) -- the constant zero, the input
    // words and the share sums -- and the launches [0, prefix_launches) that produce the sums are the same
    // for every lambda (lambda enters after them, linear.oc:52-57): garbled once, shared by all circuits
    uint32_t shared_end, prefix_launches;
    uint64_t prefix_steps;

    // ---- builder state
    size_t merge_hint = 1;       // this program will be replicated this many times (replicate_program): the dot products of
                                 // ONE circuit then need only 1 / merge_hint of the records that fill the chip
    uint64_t cap_steps;          // split launches above this many steps
    uint64_t step_cursor;
    std::map<std::pair<uint32_t, uint32_t>, std::pair<uint64_t, uint64_t>> cost_cache;
    std::map<std::pair<uint32_t, uint32_t>, uint64_t> xor_cache;
    bool open;

    Program() : w(64), p(56), d(0), T(0), nshares(0), n_words(1), n_reveal(0), in_base(0), rv_beta(0),
                rv_trace(~0u), rv_ab(~0u), total_steps(0), total_gates(0), max_launch_steps(0),
                replicas(1), word_stride(0), reveal_stride(0), lam_rec(~0u), shared_end(1), prefix_launches(0),
                prefix_steps(0), cap_steps(kDefaultCapSteps), step_cursor(0), open(false) {}

    uint32_t alloc(size_t n) { uint32_t r = n_words; n_words += (uint32_t)n; return r; }
    uint32_t alloc_reveal(size_t n) { uint32_t r = n_reveal; n_reveal += (uint32_t)n; return r; }

    // (op, cnt) of the record costed last and its figures: consecutive records are mostly of one kind, and a merged sweep
    // emits millions of them (two map lookups per record were most of the 0.2-0.3 s a 64-circuit program took to build)
    uint32_t memo_op = ~0u, memo_cnt = 0;
    uint64_t memo_steps = 0, memo_gates = 0, memo_xors = 0;
    void cost(const Rec &r, uint64_t &steps, uint64_t &gates) {
        // cost depends on (op, cnt) only -- for OP_IDIVC (cnt is 1) on the divisor: its multiplier's set bits are the steps
        const uint32_t cnt = r.op == OP_IDIVC ? r.c : r.cnt;
        if (r.op == memo_op && cnt == memo_cnt) { steps = memo_steps; gates = memo_gates; return; }
        std::pair<uint32_t, uint32_t> key(r.op, cnt);
        auto it = cost_cache.find(key);
        if (it == cost_cache.end()) {
            uint64_t s, g, x = 0;
            rec_cost(r, w, p, s, g, &x);
            it = cost_cache.insert(std::make_pair(key, std::make_pair(s, g))).first;
            xor_cache[key] = x;
        }
        steps = it->second.first;
        gates = it->second.second;
        memo_op = r.op; memo_cnt = cnt; memo_steps = steps; memo_gates = gates; memo_xors = xor_cache[key];
    }

    void new_launch() { open = false; }

    void emit(Rec r) {
        uint64_t s, g;
        cost(r, s, g);
        bool mac = (r.op == OP_MAC || r.op == OP_MAC2 || r.op == OP_MACK);
        const bool mack = r.op == OP_MACK;
        // the table cap is for the MAC launches (fewer, larger launches: kDefaultCapSteps); every other launch is cut at
        // kGenericCapSteps as in rounds 1-4 -- the input division of an 8-circuit sweep block is 18.7 M steps, and as ONE
        // launch it would set the size of the block's table ring (36 GiB instead of 18)
        const uint64_t cap_here = mac || cap_steps < kGenericCapSteps ? cap_steps : kGenericCapSteps;
        if (!open || launches.back().steps + s > cap_here || launches.back().mac_only != mac || launches.back().mack != mack) {
            Launch L;
            L.first_rec = (uint32_t)recs.size();
            L.nrec = 0;
            L.step0 = step_cursor;
            L.steps = 0;
            L.gates = 0;
            L.mac_only = mac;
            L.mack = mack;
            launches.push_back(L);
            open = true;
        }
        Launch &L = launches.back();
        r.step0 = step_cursor;
        recs.push_back(r);
        L.nrec++;
        L.steps += s;
        L.gates += g;
        step_cursor += s;
        total_steps += s;
        total_gates += g;
        total_xors += memo_xors;                       // (cost(r) above left r's figures in the memo)
        if (L.steps > max_launch_steps) max_launch_steps = L.steps;
    }

    static Rec mk(uint32_t op, uint32_t dst, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0, uint32_t cnt = 1,
                  int32_t sa = 1, int32_t sb = 1) {
        Rec r;
        r.op = op; r.cnt = cnt; r.dst = dst; r.a = a; r.b = b; r.c = c; r.sa = sa; r.sb = sb; r.step0 = 0;
        return r;
    }

    // dst = max over n words at src (stride 1) and the constant-zero word; tree of OP_MAX
    void max_tree(uint32_t dst, uint32_t src, size_t n, uint32_t scratch) {
        const size_t fan = 8;
        uint32_t cur = src;
        size_t cnt = n;
        uint32_t buf = scratch;
        new_launch();
        while (cnt > fan) {
            size_t groups = (cnt + fan - 1) / fan;
            for (size_t g = 0; g < groups; g++) {
                size_t len = (g + 1) * fan <= cnt ? fan : cnt - g * fan;
                emit(mk(OP_MAX, buf + (uint32_t)g, cur + (uint32_t)(g * fan), 0, 0, (uint32_t)len));
            }
            new_launch();
            cur = buf;
            buf += (uint32_t)groups;
            cnt = groups;
        }
        // the initial ng = 0 (cgd.oc:98-101,140): at w = 64 the compare is unsigned (obig_cmp) and max(x, 0) is x -- nothing
        // to fold in; at w = 32 it is signed and a magnitude of INT_MIN loses against the zero: one more record
        if (w == 64) {
            emit(mk(OP_MAX, dst, cur, 0, 0, (uint32_t)cnt));
            new_launch();
            return;
        }
        uint32_t tmp = buf;
        emit(mk(OP_MAX, tmp, cur, 0, 0, (uint32_t)cnt));
        new_launch();
        emit(mk(OP_MAX, dst, tmp, 0, 0, 2, -(int32_t)tmp));  // words[tmp], words[0] (= const zero)
        new_launch();
    }
    static size_t max_tree_scratch(size_t n) { return n / 4 + 16; }

    // dot products in carry-save form, chunked so that a launch has enough waves.
    // result words: dst[i] = base[i] - sum_k A[i][k]*B[k]  (subtract) or  = sum (no base)
    // kdelta != 0 (64-bit only): the products go through the Karatsuba circuit (OP_MACK); hdiff of every operand
    // word of the job lies kdelta words above it (the caller has emitted the OP_HDIFF records)
    struct DotJob { uint32_t dst, base, a, b; uint32_t len; bool has_base; uint32_t kdelta = 0; };
    // MAC records of a batch of dot products for `chunk` products per record (two chunks per record
    // when w == 32); fills the partial-word bookkeeping of every job
    void dots_records(const std::vector<DotJob> &jobs, uint32_t scratch, size_t chunk, std::vector<Rec> &out,
                      std::vector<std::pair<uint32_t, uint32_t>> &parts, bool kara_ok = true) const {
        out.clear();
        parts.assign(jobs.size(), std::make_pair(0u, 0u));   // (first partial word, count of words)
        uint32_t cur = scratch;
        for (size_t i = 0; i < jobs.size(); i++) {
            const DotJob &J = jobs[i];
            parts[i].first = cur;
            uint32_t nparts = 0;
            for (uint32_t k0 = 0; k0 < J.len;) {
                uint32_t left = J.len - k0;
                if (w == 32 && left >= 2) {
                    // two chunks of `len` products side by side in one wave (lanes 0..31 / 32..63)
                    uint32_t len = left / 2 < chunk ? left / 2 : (uint32_t)chunk;
                    out.push_back(mk(OP_MAC2, cur, J.a + k0, J.b + k0, 0, len));
                    cur += 4;
                    nparts += 4;
                    k0 += 2 * len;
                } else {
                    // ceil(len / chunk) records of (nearly) equal length rather than full chunks and a remainder: the
                    // records of a launch run in lock step, round by round.  Karatsuba records take their products two
                    // at a time: even lengths (a single leftover product would be paired with the zero word)
                    const uint32_t nrec_job = (uint32_t)((left + chunk - 1) / chunk);      // (w == 32: `left` is the odd last product)
                    const bool kara = J.kdelta && w == 64 && kara_ok;
                    const uint32_t unit = kara ? 2u : 1u, units = (left + unit - 1) / unit;
                    for (uint32_t q = 0; q < nrec_job && k0 < J.len; q++) {
                        uint32_t un = units / nrec_job + (q < units % nrec_job ? 1u : 0u);
                        uint32_t len = un * unit;
                        if (len > J.len - k0) len = J.len - k0;
                        if (!len) continue;
                        out.push_back(mk(kara ? OP_MACK : OP_MAC, cur, J.a + k0, J.b + k0, kara ? J.kdelta : 0, len));
                        cur += 2;
                        nparts += 2;
                        k0 += len;
                    }
                }
            }
            parts[i].second = nparts;
        }
    }
    // Launch shaping.  The MAC kernels run one workgroup per CU with 16 (garbler) or 12 (evaluator)
    // records each, and every record of a batch takes the same time, so a launch proceeds in rounds
    // of 4096 / 3072 records: a launch of 10 400 records costs three garbler rounds but fills 2.55.
    // Among the chunk sizes down to half the default and the launch counts that respect the table
    // cap, pick the pair with the least rounds x steps, and split the records evenly.
    // (Round 4: the callers' record targets -- kMvRecords64/32, kFactRecords below -- now ask for records so short that a big
    // launch is tens of rounds and the round model only decides between neighbouring chunk sizes; it still matters for
    // mid-size batches and for the launch count of a merged sweep.)
    static const size_t kRoundRecs = 12288;          // lcm(256 x 16, 256 x 12)
    static double round_cost(size_t per, size_t quantum) {   // a workgroup is the unit: a partial round costs a round
        return (double)((per + quantum - 1) / quantum);
    }
    // relative time of a batch of MAC records run as `launches` equal launches, longest records first: a round takes as
    // long as its first (longest) record; garbler rounds of 4096 records weigh 64 (16 waves per CU, 4 AES per gate),
    // evaluator rounds of 3072 weigh 24 (12 waves, 2 AES).  rep: every record stands for `rep` equal ones (the circuits
    // of a merged sweep, replicate_program).
    static double shaped_cost(const std::vector<uint64_t> &steps_desc, size_t rep, size_t launches) {
        const size_t N = steps_desc.size() * rep, per = (N + launches - 1) / launches;
        double t = 3e2 * (double)launches;
        for (size_t lo = 0; lo < N; lo += per) {
            const size_t hi = lo + per < N ? lo + per : N;
            for (size_t i = lo; i < hi; i += 4096) t += 64.0 * (double)steps_desc[i / rep];
            for (size_t i = lo; i < hi; i += 3072) t += 24.0 * (double)steps_desc[i / rep];
        }
        return t;
    }
    // kara_min: Karatsuba records (jobs with kdelta) where the batch has more than kara_min products; 0 = where the
    // default chunk holds two products
    void dots(const std::vector<DotJob> &jobs, uint32_t scratch, size_t target_waves, size_t kara_min = 0) {
        size_t total = 0;
        bool any_kara = false;
        for (size_t i = 0; i < jobs.size(); i++) { total += jobs[i].len; any_kara = any_kara || (jobs[i].kdelta && w == 64); }
        if (total == 0) return;
        size_t c0 = dots_chunk(total, target_waves), clo = dots_chunk_low(total, target_waves);
        // Karatsuba records take their products in pairs: only where the batch is large enough for two products per record
        // (the early and late columns of a factorisation are not: a lone product paired with the zero word costs 220 steps)
        const bool kara_ok = kara_min ? total > kara_min : c0 >= 2;
        if (any_kara && kara_ok) { if (c0 < 2) c0 = 2; if (clo < 2) clo = 2; }
        // chunk sizes from half to one and a half times the default (larger chunks: fewer, longer records -- and fewer
        // partial sums to merge); the scratch for the partial sums is sized for the smallest chunk
        size_t chi = c0 + c0 / 2;
        {
            uint64_t s1, g1;
            cost(mk(OP_MAC, 0, 0, 0, 0, 1), s1, g1);
            const size_t by_slot = (size_t)(cap_steps / ((uint64_t)kMinRecsPerLaunch * s1));
            if (chi > by_slot) chi = by_slot > c0 ? by_slot : c0;
        }
        const size_t rep = merge_hint ? merge_hint : 1;
        std::vector<Rec> recs_best, recs_try;
        std::vector<std::pair<uint32_t, uint32_t>> parts, parts_try;
        std::vector<uint64_t> sdesc;
        double best = -1.0;
        size_t best_launches = 1;
        for (size_t c = chi; c >= clo; c--) {
            dots_records(jobs, scratch, c, recs_try, parts_try, kara_ok);
            uint64_t steps = 0, smax = 0;
            sdesc.resize(recs_try.size());
            for (size_t i = 0; i < recs_try.size(); i++) {
                uint64_t s1, g1;
                cost(recs_try[i], s1, g1);
                steps += s1;
                if (s1 > smax) smax = s1;
                sdesc[i] = s1;
            }
            std::sort(sdesc.begin(), sdesc.end(), [](uint64_t x, uint64_t y) { return x > y; });
            const size_t R = recs_try.size();
            const size_t lmin = (size_t)((steps + cap_steps - 1) / cap_steps);
            for (size_t L = lmin ? lmin : 1; L <= lmin + 3; L++) {
                size_t per = (R + L - 1) / L;
                if ((uint64_t)per * smax > cap_steps) continue;
                // a merged sweep is ONE batch of rep x R records (replicate_program cuts it by the table cap afterwards)
                const double c_est = shaped_cost(sdesc, rep, rep > 1 ? 1 : L);
                if (best < 0 || c_est < best) { best = c_est; best_launches = L; recs_best = recs_try; parts = parts_try; }
                if (rep > 1) break;
            }
            if (c == 1) break;
        }
        if (best < 0) {   // cannot happen (L = lmin + 3 always fits); keep the default shape
            dots_records(jobs, scratch, c0, recs_best, parts, kara_ok);
            best_launches = 0;
        }
        {   // the partial sums of this call must fit the scratch its caller allocated
            uint32_t end = scratch;
            for (size_t i = 0; i < parts.size(); i++) if (parts[i].first + parts[i].second > end) end = parts[i].first + parts[i].second;
            auto cap = dots_caps.find(scratch);
            if (cap == dots_caps.end() || (size_t)(end - scratch) > cap->second) overflow = true;
        }
        new_launch();
        // longest records first: a round of the chip then holds records of one length (the records of a launch are
        // independent, so their order is free)
        std::stable_sort(recs_best.begin(), recs_best.end(), [](const Rec &x, const Rec &y) { return x.cnt > y.cnt; });
        const size_t R = recs_best.size();
        const size_t per = best_launches ? (R + best_launches - 1) / best_launches : R;
        for (size_t i = 0; i < R; i++) {
            if (i && per && i % per == 0) new_launch();
            emit(recs_best[i]);
        }
        new_launch();
        // Merging the partial sums of a job is a chain of carry-save steps as long as the job has parts (two words per
        // piece).  Where few jobs run at once the chain IS the run time of the merge launch (Cholesky at d = 100: 198
        // merge launches of up to 198 + 14 dependent steps, 8 % of the solve), so long merges go in two levels: groups of
        // about sqrt(parts) words are resolved side by side, then the group sums are merged (sums mod 2^w: any grouping
        // gives the same word).  Many jobs at once (a merged lambda sweep) are throughput-bound and keep the single
        // level: the extra additions would cost more than the shorter chain saves.
        size_t maxparts = 0;
        for (size_t i = 0; i < jobs.size(); i++) if (jobs[i].len && parts[i].second > maxparts) maxparts = parts[i].second;
        if (jobs.size() <= kSumTreeMaxJobs && maxparts >= kSumTreeMinParts) {
            std::vector<uint32_t> gsz(jobs.size(), 0), gcnt(jobs.size(), 0);
            size_t need = 0;
            for (size_t i = 0; i < jobs.size(); i++) {
                if (jobs[i].len == 0) continue;
                uint32_t cnt = parts[i].second, m = 1;
                while ((uint64_t)m * m < cnt) m++;
                gsz[i] = m; gcnt[i] = (cnt + m - 1) / m;
                need += gcnt[i];
            }
            if (need > sum_tree_cap) { sum_tree_cap = need + need / 2 + 16; sum_tree_base = alloc(sum_tree_cap); }
            uint32_t off = 0;
            std::vector<uint32_t> first(jobs.size(), 0);
            for (size_t i = 0; i < jobs.size(); i++) {
                if (jobs[i].len == 0) continue;
                first[i] = sum_tree_base + off;
                for (uint32_t g = 0; g < gcnt[i]; g++) {
                    uint32_t lo = g * gsz[i], n = parts[i].second - lo < gsz[i] ? parts[i].second - lo : gsz[i];
                    emit(mk(OP_SUM, first[i] + g, parts[i].first + lo, 0, 0, n));
                }
                off += gcnt[i];
            }
            new_launch();
            for (size_t i = 0; i < jobs.size(); i++) {
                const DotJob &J = jobs[i];
                if (J.len == 0) continue;
                if (J.has_base) emit(mk(OP_SUBSUM, J.dst, first[i], 0, J.base, gcnt[i]));
                else emit(mk(OP_SUM, J.dst, first[i], 0, 0, gcnt[i]));
            }
            new_launch();
            return;
        }
        for (size_t i = 0; i < jobs.size(); i++) {
            const DotJob &J = jobs[i];
            if (J.len == 0) continue;
            if (J.has_base) emit(mk(OP_SUBSUM, J.dst, parts[i].first, 0, J.base, parts[i].second));
            else emit(mk(OP_SUM, J.dst, parts[i].first, 0, 0, parts[i].second));
        }
        new_launch();
    }
    static constexpr size_t kSumTreeMaxJobs = 1024, kSumTreeMinParts = 32;
    uint32_t sum_tree_base = 0;          // scratch words of the first merge level (grow-only, shared by all dots() calls)
    size_t sum_tree_cap = 0;
    // products per OP_MAC record: enough records to fill the chip (target_waves), and
    // short enough that one table slot (cap_steps gate steps) still holds >= kMinRecsPerLaunch
    // records -- a launch with fewer waves than the GPU has wave slots idles most CUs
    size_t dots_chunk(size_t total, size_t target_waves) {
        size_t chunk = (total + target_waves - 1) / target_waves;
        uint64_t s1, g1;
        cost(mk(OP_MAC, 0, 0, 0, 0, 1), s1, g1);      // (an upper bound for OP_MACK records as well)
        size_t by_slot = (size_t)(cap_steps / ((uint64_t)kMinRecsPerLaunch * s1));
        if (chunk > by_slot) chunk = by_slot;
        if (chunk < 1) chunk = 1;
        return chunk;
    }
    // smallest chunk the launch shaping in dots() may pick (sizes the scratch for the partial sums)
    size_t dots_chunk_low(size_t total, size_t target_waves) {
        size_t c = dots_chunk(total, target_waves) / 2;
        return c < 1 ? 1 : c;
    }
    // Scratch words for the partial sums of ONE dots() call (two words per record, four per dual 32-bit record), whatever
    // its batch: with c0 = ceil(total / target_waves) the smallest chunk tried is max(1, c0 / 2) >= c0 / 3, so a call makes
    // at most 3 * target_waves + njobs records (total_products: an upper bound on the products of any one call).  (Round 2 sized this from the total of the LARGEST call at ITS smallest
    // chunk; a smaller call has its own, relatively smaller, smallest chunk -- c0 = 4 gives 2 -- and with the larger table
    // cap of round 3 the 32-bit Cholesky at d = 250 ran 4 000 words past the end.  build_program now also verifies that
    // every record stays inside the word file: Program::ranges_ok.)
    size_t dots_scratch(size_t total_products, size_t njobs, size_t target_waves) {
        // ... unless the table cap, not the target, bounds the chunk (dots_chunk: by_slot): then a call makes up to
        // total / max(1, by_slot / 2) records
        uint64_t s1, g1;
        cost(mk(OP_MAC, 0, 0, 0, 0, 1), s1, g1);
        const size_t by_slot = (size_t)(cap_steps / ((uint64_t)kMinRecsPerLaunch * s1));
        const size_t lo = by_slot / 2 ? by_slot / 2 : 1;
        size_t recs = 3 * target_waves;
        if (total_products / lo + 1 > recs) recs = total_products / lo + 1;
        if (recs > total_products) recs = total_products;                 // (never more records than products)
        return 2 * (recs + njobs + 2) + 4 * njobs + 16;   // + tails of dual 32-bit records
    }
    // allocate the scratch of a series of dots() calls and remember its size: dots() checks every call against it
    // (a call that does not fit marks the program `overflow`, which the engine refuses to run)
    std::map<uint32_t, size_t> dots_caps;
    bool overflow = false;
    uint32_t alloc_dots(size_t total_products, size_t njobs, size_t target_waves, size_t extra = 0) {
        const size_t n = dots_scratch(total_products, njobs, target_waves) + extra;
        const uint32_t base = alloc(n);
        dots_caps[base] = n;
        return base;
    }
    // every word a record touches lies inside the word file (checked once per built program)
    bool ranges_ok() const {
        if (overflow) return false;
        for (size_t i = 0; i < recs.size(); i++) {
            const Rec &r = recs[i];
            uint64_t hi = 0;
            auto upd = [&hi](uint64_t x) { if (x > hi) hi = x; };
            const uint64_t n = r.cnt ? r.cnt : 1;
            switch (r.op) {
            case OP_MAC: upd(r.dst + 1); upd((uint64_t)((int64_t)r.a + (int64_t)(n - 1) * r.sa)); upd((uint64_t)((int64_t)r.b + (int64_t)(n - 1) * r.sb)); break;
            case OP_MAC2: upd(r.dst + 3); upd((uint64_t)((int64_t)r.a + (int64_t)(2 * n - 1) * r.sa)); upd((uint64_t)((int64_t)r.b + (int64_t)(2 * n - 1) * r.sb)); break;
            case OP_MACK: upd(r.dst + 1); upd((uint64_t)((int64_t)r.a + (int64_t)(n - 1) * r.sa) + r.c); upd((uint64_t)((int64_t)r.b + (int64_t)(n - 1) * r.sb) + r.c); break;
            case OP_SUM: case OP_SUBSUM: case OP_MAX: upd(r.dst); upd(r.a); upd((uint64_t)((int64_t)r.a + (int64_t)(n - 1) * r.sa)); if (r.op == OP_SUBSUM) upd(r.c); break;
            case OP_IPMAC: upd(r.dst + 3); upd(r.a + n - 1); upd(r.b + n - 1); break;
            case OP_IPFIN: case OP_IPMERGE: upd(r.dst + (r.op == OP_IPMERGE ? 3 : 0)); upd(r.a + 4 * n - 1); break;
            case OP_CONST: upd(r.dst); break;
            case OP_REVEAL: upd(r.a); break;
            case OP_IDIVC: case OP_COPY: case OP_ABS: case OP_SQRT: case OP_HDIFF: upd(r.dst); upd(r.a); break;
            case OP_MULSUB: upd(r.dst); upd(r.a); upd(r.b); upd(r.c); if (r.cnt >= 2) upd((uint32_t)(r.dst + (uint32_t)r.sa)); break;   // the second store's offset wraps in 32 bits, as in exec_record
            case OP_DIVB: upd(r.dst); upd(r.a); upd(r.b); break;
            case OP_DIV: case OP_MUL: upd(r.dst); upd(r.a); upd(r.b); if (r.op == OP_DIV) upd(r.c); if (r.cnt == 2) upd((uint32_t)(r.dst + (uint32_t)r.sa)); break;
            default: upd(r.dst); upd(r.a); upd(r.b); break;
            }
            if (hi >= n_words) return false;
        }
        return true;
    }

    // wide inner products (fixed.oc:124-147), several independent ones level-synchronously:
    // one product per record, then a fan-in-4 merge tree of carry-save accumulators
    struct IpJob { uint32_t dst, a, b; };
    void inners(const std::vector<IpJob> &jobs, size_t n, uint32_t scratch) {
        const size_t fan = 4;
        const size_t per = inner_scratch(n) / 1;   // words reserved per job
        new_launch();
        for (size_t j = 0; j < jobs.size(); j++) {
            uint32_t base = scratch + (uint32_t)(j * per);
            for (size_t k = 0; k < n; k++)
                emit(mk(OP_IPMAC, base + (uint32_t)(4 * k), jobs[j].a + (uint32_t)k, jobs[j].b + (uint32_t)k, 0, 1));
        }
        new_launch();
        size_t cnt = n;
        uint32_t off_cur = 0, off_next = (uint32_t)(4 * n);
        while (cnt > fan) {
            size_t groups = (cnt + fan - 1) / fan;
            for (size_t j = 0; j < jobs.size(); j++) {
                uint32_t base = scratch + (uint32_t)(j * per);
                for (size_t g = 0; g < groups; g++) {
                    size_t len = (g + 1) * fan <= cnt ? fan : cnt - g * fan;
                    emit(mk(OP_IPMERGE, base + off_next + (uint32_t)(4 * g), base + off_cur + (uint32_t)(4 * g * fan), 0, 0,
                            (uint32_t)len));
                }
            }
            new_launch();
            off_cur = off_next;
            off_next += (uint32_t)(4 * groups);
            cnt = groups;
        }
        for (size_t j = 0; j < jobs.size(); j++) {
            uint32_t base = scratch + (uint32_t)(j * per);
            emit(mk(OP_IPFIN, jobs[j].dst, base + off_cur, 0, 0, (uint32_t)cnt));
        }
        new_launch();
    }
    void inner(uint32_t dst, uint32_t a, uint32_t b, size_t n, uint32_t scratch) {
        std::vector<IpJob> jobs(1);
        IpJob J = {dst, a, b};
        jobs[0] = J;
        inners(jobs, n, scratch);
    }
    static size_t inner_scratch(size_t n) { return 4 * n + 4 * (n / 3 + 8) + 16; }   // per job
};

// Records per big multiply-accumulate launch.  Rounds 1-3 shaped these launches to whole rounds of the chip (12 288 records of
// ~21 products: three garbler rounds of 4 096 waves, four evaluator rounds of 3 072) -- but records of one length retire in lock
// step, and the OTHER chain's dependent small launches (sums, inner products, the scalar dividers between two matrix-vector
// products) then get CUs only at a round boundary: one launch of that chain per ~11 ms round, so the chain ran past the MAC
// kernel it was meant to hide behind and the next matrix-vector product started 4-7 ms late, every iteration
// (kernel timelines: profiles/r4_timeline_d500_cgd3_*.txt).  Short records (two products = one Karatsuba pair, 220 gate
// steps, ~1.5 ms) turn workgroups over continuously: the chain finishes in half the MAC kernel's time, the MAC kernels
// themselves lose nothing (d = 500 CGD-15: 1.99 -> 1.81 s, d = 300: 0.78 -> 0.71 s; scripts/exp/shape_ab*.sh).  The price is
// more partial sums to merge (+1 % gate steps at d = 500).  kTargetWaves still decides WHERE Karatsuba records are used
// (batches of more than that many products), so programs of small systems are what they were.
// (End of round 5: 8 192 instead of 12 288 -- d = 91 ... 110 get Karatsuba records too -- now that a Karatsuba launch of up to
// three rounds picks its waves per workgroup, gc_mack_waves: d = 100 CGD-15 0.128 -> 0.122 s, scripts/exp/kara_small_ab.sh;
// with sixteen-wave workgroups the 5 000 pairs of d = 100 were rounds of 16 + 3.5 waves and 3 % SLOWER than 10 000 plain records.)
static const size_t kTargetWaves = 8192;
static const size_t kMvRecords64 = 131072;       // matrix-vector products of CGD, 64-bit (chunk floor: one Karatsuba pair)
static const size_t kMvRecords32 = 65536;        // ... 32-bit (two-chunk OP_MAC2 records; 131 072 costs 5 % more steps)
static const size_t kFactRecords = 65536;        // a column step of Cholesky / LDL^T (d = 500: 12.6 -> 12.0 s)
static inline size_t x_fact_waves() { return kFactRecords; }     // records per column step of the factorisations

// Build the whole phase-2 program.
//   normalize = 1: data-provider path (linear.oc:52-65): diag += lambda, off-diag and b divided by d
//   normalize = 0: two-party benchmark path (linear.oc:96-135): a = in1 + in2, nothing else
//   reveal_ab: debug reveal of a and b (linear.oc:68-84)
// Karatsuba products in the matrix-vector launches of CGD (64-bit; Circ::mack2).  Process-wide switch for A/B runs
// (lgc_set_karatsuba); garbler and evaluator must agree, as on everything else that shapes the program.
inline int &program_karatsuba() { static int on = 1; return on; }
// OP_DIVB (p + 1 quotient bits) for CGD's g / max|g| at w = 64
inline int program_bounded_div() { return 1; }

inline void build_program(Program &P, int alg, size_t d, int w, int p, int iters, size_t nshares,
                          int normalize, uint64_t lambda_fixed, int reveal_ab, int trace) {
    P.w = w; P.p = p; P.d = d; P.nshares = nshares;
    const size_t T = d * (d + 1) / 2;
    P.T = T;
    const uint32_t D = (uint32_t)d;
    // word 0 is the constant zero (the word file starts zeroed on both sides)
    P.in_base = P.alloc(nshares * (T + d));
    if (alg == ALG_DIMCHECK) {
        // "check if inputs have equal dimensions" (src/linear.oc:109-114): the first word of either party's input is its d;
        // one comparison, revealed.  d = 1, two shares: a program that does not depend on what it checks
        const uint32_t eq = P.alloc(1);
        P.new_launch();
        P.emit(Program::mk(OP_EQ, eq, P.in_base, P.in_base + (uint32_t)(T + d)));
        P.new_launch();
        P.rv_beta = P.alloc_reveal(d);
        P.emit(Program::mk(OP_REVEAL, P.rv_beta, eq));
        P.new_launch();
        return;
    }
    const uint32_t S_first = normalize ? P.alloc(T + d) : 0;   // share sums (see below): directly after the inputs
    const uint32_t M = P.alloc(d * d);       // full symmetric storage, M[i*d+j] == M[j*d+i]
    const uint32_t bv = P.alloc(d);
    auto Mi = [&](size_t i, size_t j) { return M + (uint32_t)(i * d + j); };
    auto idx = [](size_t i, size_t j) { return (uint32_t)(i * (i + 1) / 2 + j); };

    // ---- a[ij] = sum of shares (linear.oc:31-49 / :116-127).  On the data-provider path the sums go to
    // their own words S (right after the inputs): everything up to here does not depend on lambda, so a
    // sweep garbles it once and every circuit of the sweep reads S (replicate_program)
    const uint32_t S = S_first;
    P.new_launch();
    for (size_t i = 0; i < d; i++)
        for (size_t j = 0; j <= i; j++)
            P.emit(Program::mk(OP_SUM, normalize ? S + idx(i, j) : Mi(i, j), P.in_base + idx(i, j), 0, 0, (uint32_t)nshares,
                               (int32_t)(T + d)));
    for (size_t i = 0; i < d; i++)
        P.emit(Program::mk(OP_SUM, normalize ? S + (uint32_t)(T + i) : bv + (uint32_t)i, P.in_base + (uint32_t)(T + i), 0, 0,
                           (uint32_t)nshares, (int32_t)(T + d)));
    P.new_launch();
    if (normalize) {
        // the division by the public normalizer (linear.oc:57-65) does not depend on lambda either: in place on the share
        // sums, still in the prefix -- a sweep divides once, not once per circuit (1.5 % of a d = 100 CGD-15 circuit)
        for (size_t i = 0; i < d; i++)
            for (size_t j = 0; j < i; j++) P.emit(idivc_rec(S + idx(i, j), S + idx(i, j), D, w));
        for (size_t i = 0; i < d; i++) P.emit(idivc_rec(S + (uint32_t)(T + i), S + (uint32_t)(T + i), D, w));
        P.new_launch();
        P.shared_end = S + (uint32_t)(T + d);
        P.prefix_launches = (uint32_t)P.launches.size();
        P.prefix_steps = P.total_steps;
        const uint32_t lam = P.alloc(1);
        P.lam_rec = (uint32_t)P.recs.size();
        P.emit(Program::mk(OP_CONST, lam, (uint32_t)lambda_fixed, (uint32_t)(lambda_fixed >> 32)));
        P.new_launch();
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_ADD, Mi(i, i), S + idx(i, i), lam));     // linear.oc:54-56
        P.new_launch();
        // the circuit's own copy of the rest, both triangles (the factorisations work in place)
        for (size_t i = 0; i < d; i++)
            for (size_t j = 0; j < i; j++) {
                P.emit(Program::mk(OP_COPY, Mi(i, j), S + idx(i, j)));
                P.emit(Program::mk(OP_COPY, Mi(j, i), S + idx(i, j)));
            }
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_COPY, bv + (uint32_t)i, S + (uint32_t)(T + i)));
        P.new_launch();
    } else {
        // mirror the lower triangle
        for (size_t i = 0; i < d; i++)
            for (size_t j = 0; j < i; j++) P.emit(Program::mk(OP_COPY, Mi(j, i), Mi(i, j)));
        P.new_launch();
    }
    if (reveal_ab) {
        P.rv_ab = P.alloc_reveal(T + d);
        for (size_t i = 0; i < d; i++)
            for (size_t j = 0; j <= i; j++) P.emit(Program::mk(OP_REVEAL, P.rv_ab + idx(i, j), Mi(i, j)));
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_REVEAL, P.rv_ab + (uint32_t)(T + i), bv + (uint32_t)i));
        P.new_launch();
    }

    if (alg == ALG_CGD) {
        const uint32_t x = P.alloc(d), g = P.alloc(d), pv = P.alloc(d), gscl = P.alloc(d), pA = P.alloc(d),
                       tabs = P.alloc(d);
        const uint32_t ng = P.alloc(1), q = P.alloc(1), gp = P.alloc(1), eta = P.alloc(1), gamma = P.alloc(1),
                       gAp = P.alloc(1);
        const uint32_t sc_max = P.alloc(Program::max_tree_scratch(d));
        const uint32_t sc_ip = P.alloc(2 * Program::inner_scratch(d));
        // records per matrix-vector product: enough to fill the chip -- together with the other circuits of a merged sweep
        size_t mv_target = w == 64 ? kMvRecords64 : kMvRecords32;
        size_t mv_waves = mv_target / (P.merge_hint ? P.merge_hint : 1);
        if (mv_waves < 2 * d) mv_waves = 2 * d < mv_target ? 2 * d : mv_target;     // at least two records per row
        size_t kara_target = kTargetWaves;
        size_t kara_min = kara_target / (P.merge_hint ? P.merge_hint : 1);          // Karatsuba products where d * d exceeds this
        if (kara_min < 2 * d) kara_min = 2 * d < kara_target ? 2 * d : kara_target;
        const uint32_t sc_dot = P.alloc_dots(d * d, d, mv_waves);
        if (trace) P.rv_trace = P.alloc_reveal((size_t)iters * (d + 4));
        // Karatsuba products for A p (w = 64): the words hdiff(M[i][j]) -- once per solve -- and hdiff(p[k]) -- once per
        // iteration -- live in a shadow of the word range [M, pv + d), kdelta words above their operands
        uint32_t kdelta = 0;
        if (w == 64 && iters > 0 && program_karatsuba() && d * d > kara_min) {   // (needs two products per record)
            kdelta = P.alloc((size_t)(pv + D - M)) - M;
            for (size_t i = 0; i < d; i++)
                for (size_t j = 0; j <= i; j++) P.emit(Program::mk(OP_HDIFF, Mi(i, j) + kdelta, Mi(i, j)));
            P.new_launch();
            for (size_t i = 0; i < d; i++)
                for (size_t j = 0; j < i; j++) P.emit(Program::mk(OP_COPY, Mi(j, i) + kdelta, Mi(i, j) + kdelta));
            P.new_launch();
        }
        // cgd.oc:96-106
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_SUB, g + (uint32_t)i, 0, bv + (uint32_t)i));
        P.new_launch();
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_ABS, tabs + (uint32_t)i, g + (uint32_t)i));
        P.max_tree(ng, tabs, d, sc_max);
        // g_i / max_j |g_j|: a quotient of at most 2^p.  At w = 64 the maximum is an UNSIGNED maximum of the very magnitudes
        // the divider forms (Circ::vabs, Circ::gt), so |g_i| <= |ng| holds for every input and the divider may skip the
        // quotient bits above p (OP_DIVB); at w = 32 the compare is signed (fixed.oc:78-88) and |INT_MIN| escapes it
        const uint32_t op_divb = (w == 64 && program_bounded_div()) ? OP_DIVB : OP_DIV;
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(op_divb, pv + (uint32_t)i, g + (uint32_t)i, ng));
        P.new_launch();
        for (int it = 0; it < iters; it++) {
            // pA = A p  (cgd.oc:119-125)
            if (kdelta && it == 0) {                 // (later iterations: the record that makes p_i forms hdiff(p_i), below)
                for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_HDIFF, pv + (uint32_t)i + kdelta, pv + (uint32_t)i));
                P.new_launch();
            }
            std::vector<Program::DotJob> jobs(d);
            for (size_t i = 0; i < d; i++) {
                Program::DotJob J = {pA + (uint32_t)i, 0, Mi(i, 0), pv, D, false, kdelta};
                jobs[i] = J;
            }
            P.dots(jobs, sc_dot, mv_waves, kara_min);
            {                                        // q = <pA,p> (:128), gp = <g,p> (:130)
                std::vector<Program::IpJob> ij(2);
                Program::IpJob j0 = {q, pA, pv}, j1 = {gp, g, pv};
                ij[0] = j0; ij[1] = j1;
                P.inners(ij, d, sc_ip);
            }
            P.emit(Program::mk(OP_DIV, eta, gp, q)); // :133
            P.new_launch();
            for (size_t i = 0; i < d; i++) {         // :141-145
                P.emit(Program::mk(OP_MULSUB, x + (uint32_t)i, pv + (uint32_t)i, eta, x + (uint32_t)i));
                // ... and |g_i| with it (cnt = 2): the maximum below starts from these
                P.emit(Program::mk(OP_MULSUB, g + (uint32_t)i, eta, pA + (uint32_t)i, g + (uint32_t)i, 2, (int32_t)(tabs - g)));
            }
            P.max_tree(ng, tabs, d, sc_max);         // :140,146-149  (opens a launch of its own)
            for (size_t i = 0; i < d; i++)           // :153-155
                P.emit(Program::mk(op_divb, gscl + (uint32_t)i, g + (uint32_t)i, ng));
            P.new_launch();
            P.inner(gAp, pA, gscl, d, sc_ip);        // :157
            P.emit(Program::mk(OP_DIV, gamma, gAp, q));  // :159
            P.new_launch();
            for (size_t i = 0; i < d; i++)           // :162-165
                P.emit(Program::mk(OP_MULSUB, pv + (uint32_t)i, pv + (uint32_t)i, gamma, gscl + (uint32_t)i, kdelta ? 3u : 1u,
                                   kdelta ? (int32_t)kdelta : 1));
            P.new_launch();
            if (trace) {                             // reveals at :167-189
                uint32_t base = P.rv_trace + (uint32_t)((size_t)it * (d + 4));
                for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_REVEAL, base + (uint32_t)i, x + (uint32_t)i));
                P.emit(Program::mk(OP_REVEAL, base + D, gamma));
                P.emit(Program::mk(OP_REVEAL, base + D + 1, eta));
                P.emit(Program::mk(OP_REVEAL, base + D + 2, q));
                P.emit(Program::mk(OP_REVEAL, base + D + 3, ng));
                P.new_launch();
            }
            P.iter_launch.push_back((uint32_t)(P.launches.size() - 1));
            P.iter_gates.push_back(P.total_gates);
        }
        P.rv_beta = P.alloc_reveal(d);
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_REVEAL, P.rv_beta + (uint32_t)i, x + (uint32_t)i));
        P.new_launch();
    } else if (alg == ALG_CHOLESKY) {
        const uint32_t y = P.alloc(d), beta = P.alloc(d);
        const uint32_t sc_dot = P.alloc_dots(d * d + d, d + 1, x_fact_waves(), 4 * d + 8);
        // Karatsuba products in the factorisation (w = 64, large d): an entry L_kj -- and y_j -- is final once column j has
        // been scaled, so its hdiff word (shadow of [M, y + d), kdelta words up) is formed in the launch that mirrors the
        // column (independent of the copies: no launch is added to the chain); columns with fewer than two products per
        // record keep the plain array (dots()).  The back substitution (one short dot product per step) is left as it is.
        uint32_t kdelta = 0;
        if (w == 64 && program_karatsuba() && (d / 2) * (d / 2 + 1) >= 2 * 4096) kdelta = P.alloc((size_t)(y + D - M)) - M;
        // cholesky.oc:51-65 (factorisation) and :68-76 (forward substitution) as ONE chain of launches: step j of
        // the forward substitution, y_j = (b_j - sum_{k<j} L_jk y_k) / L_jj, needs row j of L (complete once
        // column j - 1 has been scaled) and y_0..y_{j-1}, so its dot product joins the dot products of column
        // j and its division joins the launch that scales column j.  Same operations on the same operands as
        // the reference's three loops (results identical); d division launches and 2d narrow launches fewer
        // on the dependent chain, which is what a small system's run time consists of.
        for (size_t j = 0; j < d; j++) {
            if (j > 0) {
                std::vector<Program::DotJob> jobs;
                for (size_t i = j; i < d; i++) {
                    Program::DotJob J = {Mi(i, j), Mi(i, j), Mi(i, 0), Mi(j, 0), (uint32_t)j, true, kdelta};
                    jobs.push_back(J);
                }
                Program::DotJob F = {bv + (uint32_t)j, bv + (uint32_t)j, Mi(j, 0), y, (uint32_t)j, true, kdelta};   // :70-73
                jobs.push_back(F);
                P.dots(jobs, sc_dot, x_fact_waves(), 4096);
            }
            P.emit(Program::mk(OP_SQRT, Mi(j, j), Mi(j, j)));
            P.new_launch();
            // the division record stores its quotient twice (L_kj and its mirror L^T_jk, read stride-1 by the back
            // substitution) and, with Karatsuba products, its half-difference word: rounds 3-4 did both in a launch of
            // their own behind the divisions -- one more dependent launch per column, each of which waits for CUs beside
            // the other role's MAC kernel of that column
            const uint32_t hc = kdelta ? 2u : 1u;
            for (size_t k = j + 1; k < d; k++) P.emit(Program::mk(OP_DIV, Mi(k, j), Mi(k, j), Mi(j, j), Mi(j, k), hc, (int32_t)kdelta));
            P.emit(Program::mk(OP_DIV, y + (uint32_t)j, bv + (uint32_t)j, Mi(j, j), 0, hc, (int32_t)kdelta));           // :75
            P.new_launch();
        }
        for (size_t ii = d; ii-- > 0;) {             // :79-87
            if (ii + 1 < d) {
                std::vector<Program::DotJob> jobs(1);
                Program::DotJob J = {y + (uint32_t)ii, y + (uint32_t)ii, Mi(ii, ii + 1), beta + (uint32_t)(ii + 1),
                                     (uint32_t)(d - 1 - ii), true};
                jobs[0] = J;
                P.dots(jobs, sc_dot, 64);
            }
            P.emit(Program::mk(OP_DIV, beta + (uint32_t)ii, y + (uint32_t)ii, Mi(ii, ii)));
            P.new_launch();
        }
        P.rv_beta = P.alloc_reveal(d);
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_REVEAL, P.rv_beta + (uint32_t)i, beta + (uint32_t)i));
        P.new_launch();
    } else {  // ALG_LDLT
        const uint32_t tv = P.alloc(d);
        const uint32_t sc_dot = P.alloc_dots(d * d + d, d + 1, x_fact_waves(), 4 * d + 8);
        // Karatsuba products as in the Cholesky lowering: hdiff of L_kj in the launch that mirrors column j, of b_j (final
        // after step j of the forward substitution) and of the products t_k = L_jk D_k in a launch of their own per column
        uint32_t kdelta = 0;
        if (w == 64 && program_karatsuba() && (d / 2) * (d / 2 + 1) >= 2 * 4096) kdelta = P.alloc((size_t)(tv + D - M)) - M;
        for (size_t j = 0; j < d; j++) {             // ldlt.oc:50-64
            if (j > 0) {
                // t_k = L_jk D_k with its half-difference word from the same record; hdiff(b_{j-1}) (final since step j - 1 of
                // the forward substitution) rides in the same launch: one launch where rounds 3-4 had two
                for (size_t k = 0; k < j; k++) P.emit(Program::mk(OP_MUL, tv + (uint32_t)k, Mi(j, k), Mi(k, k), 0, kdelta ? 2u : 1u, (int32_t)kdelta));
                if (kdelta) P.emit(Program::mk(OP_HDIFF, bv + (uint32_t)(j - 1) + kdelta, bv + (uint32_t)(j - 1)));
                P.new_launch();
                std::vector<Program::DotJob> jobs;
                for (size_t i = j; i < d; i++) {
                    Program::DotJob J = {Mi(i, j), Mi(i, j), Mi(i, 0), tv, (uint32_t)j, true, kdelta};
                    jobs.push_back(J);
                }
                // step j of the forward substitution (:67-73), b_j -= sum_{k<j} L_jk b_k, needs row j of L (complete once
                // column j - 1 has been scaled) and b_0 .. b_{j-1}: it joins the dot products of column j instead of
                // forming a chain of d - 1 launch pairs of its own after the factorisation (as in the Cholesky lowering)
                Program::DotJob F = {bv + (uint32_t)j, bv + (uint32_t)j, Mi(j, 0), bv, (uint32_t)j, true, kdelta};
                jobs.push_back(F);
                P.dots(jobs, sc_dot, x_fact_waves(), 4096);
            }
            for (size_t k = j + 1; k < d; k++) P.emit(Program::mk(OP_DIV, Mi(k, j), Mi(k, j), Mi(j, j), Mi(j, k), kdelta ? 2u : 1u, (int32_t)kdelta));
            P.new_launch();
        }
        for (size_t i = 0; i < d; i++)               // :76-79
            P.emit(Program::mk(OP_DIV, bv + (uint32_t)i, bv + (uint32_t)i, Mi(i, i)));
        P.new_launch();
        for (size_t ii = d; ii-- > 0;) {             // :82-90
            if (ii + 1 < d) {
                std::vector<Program::DotJob> jobs(1);
                Program::DotJob J = {bv + (uint32_t)ii, bv + (uint32_t)ii, Mi(ii, ii + 1), bv + (uint32_t)(ii + 1),
                                     (uint32_t)(d - 1 - ii), true};
                jobs[0] = J;
                P.dots(jobs, sc_dot, 64);
            }
        }
        P.rv_beta = P.alloc_reveal(d);
        for (size_t i = 0; i < d; i++) P.emit(Program::mk(OP_REVEAL, P.rv_beta + (uint32_t)i, bv + (uint32_t)i));
        P.new_launch();
    }
}


// `count` circuits of the per-lambda sweep in one program.  lambda is a public constant
// added to the diagonal AFTER the shares are summed (linear.oc:52-57), so the input labels and the garbled
// share summation -- P0's shared prefix -- exist once and every circuit starts from the same sums: the
// data providers run ONE label OT whatever the number of lambdas.  Circuit t uses words x + t * word_stride
// for x >= shared_end and decode slots r + t * reveal_stride, and differs only in the OP_CONST record that
// holds lambda.  The records of all circuits of one launch of P0 share a launch, so the dependent chains
// (dividers, reveals) of different circuits fill the GPU together.
// first_copy: index of this program's first circuit in the whole sweep.  Ranks of a multi-GPU sweep share
// the prefix -- hence the garbler's offset R -- so their gate ids must not collide.  The number of gate steps a
// circuit lowers to depends on the size of the block it is merged into (merge_hint shapes the dot-product
// records), so the ranges are laid out on a CANONICAL stride that no lowering reaches: circuit k of the sweep
// owns gate steps inside [prefix + k * kSweepCircuitStride, prefix + (k + 1) * kSweepCircuitStride) on whichever
// rank and in whichever block it runs -- blocks of different sizes cannot overlap (round 3 used the block's own
// per-circuit count as the stride; two blocks whose sizes differed by one then shared gate ids under one R).
// Step numbers only enter the hash tweaks (64 * step + lane < 2^59 with at most 65536 circuits) and, as differences
// within a launch, the table rows.  Returns false when a circuit does not fit its stride.
static const uint64_t kSweepCircuitStride = 1ull << 36;
inline bool replicate_program(Program &P, const Program &P0, size_t count, const uint64_t *lambda_fixed, size_t first_copy = 0) {
    P.w = P0.w; P.p = P0.p; P.d = P0.d; P.T = P0.T; P.nshares = P0.nshares;
    P.cap_steps = P0.cap_steps;
    P.shared_end = P0.shared_end;
    P.word_stride = P0.n_words - P0.shared_end;
    P.reveal_stride = P0.n_reveal;
    P.replicas = (uint32_t)count;
    P.n_words = P0.shared_end + (uint32_t)count * P.word_stride;
    P.n_reveal = (uint32_t)count * P.reveal_stride;
    P.in_base = P0.in_base; P.rv_beta = P0.rv_beta; P.rv_trace = P0.rv_trace; P.rv_ab = P0.rv_ab;
    P.lam_rec = ~0u;
    if (P0.prefix_steps >= kSweepCircuitStride || P0.total_steps - P0.prefix_steps > kSweepCircuitStride || first_copy + count > 65536)
        return false;
    size_t next_iter = 0;
    const uint32_t shared_end = P0.shared_end;
    {   // 6.4 M records (257 MB) for 64 circuits of d = 100 CGD-15: grown by doubling, the vector copied itself twice over
        size_t npre = 0;
        for (size_t li = 0; li < P0.prefix_launches && li < P0.launches.size(); li++) npre += P0.launches[li].nrec;
        P.recs.reserve(npre + (P0.recs.size() - npre) * count);
    }
    for (size_t li = 0; li < P0.launches.size(); li++) {
        const Launch &L = P0.launches[li];
        const bool prefix = li < P0.prefix_launches;
        if (li == P0.prefix_launches) {
            P.new_launch();
            P.prefix_launches = (uint32_t)P.launches.size();
            P.prefix_steps = P.total_steps;
            P.step_cursor = P0.prefix_steps + (uint64_t)first_copy * kSweepCircuitStride;
        }
        // a merged launch that exceeds the table cap is cut into EQUAL pieces (a ragged last piece of a MAC launch would
        // be a launch of a few hundred records: most of the chip idle, or the wrong kernel altogether)
        const uint64_t cap_keep = P.cap_steps;
        const uint64_t cap_l = L.mac_only || cap_keep < kGenericCapSteps ? cap_keep : kGenericCapSteps;   // (as in Program::emit)
        if (!prefix && L.nrec) {
            const uint64_t tot = L.steps * (uint64_t)count, pieces = (tot + cap_l - 1) / cap_l;
            uint64_t smax = 0;
            for (uint32_t k = 0; k < L.nrec; k++) { uint64_t s1, g1; P.cost(P0.recs[L.first_rec + k], s1, g1); if (s1 > smax) smax = s1; }
            uint64_t best_pieces = pieces;
            if (L.mac_only && pieces >= 1 && (uint64_t)L.nrec * count >= 2 * 4096) {
                // MAC launches run in whole rounds of the chip (dots()): among a few piece counts take the cheapest
                double best = -1.0;
                const uint64_t R = (uint64_t)L.nrec * count;
                for (uint64_t q = pieces; q <= pieces + 3; q++) {
                    const size_t per = (size_t)((R + q - 1) / q);
                    if ((uint64_t)per * smax > cap_l) continue;
                    const double c_est = (double)q * ((double)smax * (64.0 * Program::round_cost(per, 4096) + 24.0 * Program::round_cost(per, 3072)) + 3e2);
                    if (best < 0 || c_est < best) { best = c_est; best_pieces = q; }
                }
            }
            if (best_pieces > 1) {
                const uint64_t soft = (tot + best_pieces - 1) / best_pieces + smax;
                if (soft < cap_l) P.cap_steps = soft;
            }
        }
        // record-major: record k of every circuit, then record k + 1 ... -- the records of a launch are independent, and
        // equal records side by side make the rounds of a MAC launch uniform (dots() sorts them by length)
        for (uint32_t k = 0; k < L.nrec; k++) {
            for (size_t t = 0; t < (prefix ? 1 : count); t++) {
                const uint32_t wo = (uint32_t)t * P.word_stride, ro = (uint32_t)t * P.reveal_stride;
                Rec r = P0.recs[L.first_rec + k];
                auto mv = [wo, shared_end](uint32_t x) { return x >= shared_end ? x + wo : x; };
                switch (r.op) {
                case OP_CONST:
                    r.dst = mv(r.dst);
                    if (L.first_rec + k == P0.lam_rec) { r.a = (uint32_t)lambda_fixed[t]; r.b = (uint32_t)(lambda_fixed[t] >> 32); }
                    break;
                case OP_IDIVC: r.dst = mv(r.dst); r.a = mv(r.a); break;          // c is an immediate
                case OP_MACK: r.dst = mv(r.dst); r.a = mv(r.a); r.b = mv(r.b); break;   // c is an offset between words of one circuit
                case OP_REVEAL: r.dst += ro; r.a = mv(r.a); break;               // dst is a decode slot
                case OP_MAX: {
                    // max_tree folds in the constant zero as the SECOND operand through a stride of -a
                    const bool to_zero = r.cnt == 2 && r.sa == -(int32_t)r.a;
                    r.dst = mv(r.dst); r.a = mv(r.a);
                    if (to_zero) r.sa = -(int32_t)r.a;
                } break;
                default: r.dst = mv(r.dst); r.a = mv(r.a); r.b = mv(r.b); r.c = mv(r.c); break;
                }
                P.emit(r);
            }
        }
        P.cap_steps = cap_keep;
        P.new_launch();
        while (next_iter < P0.iter_launch.size() && P0.iter_launch[next_iter] == li) {
            P.iter_launch.push_back((uint32_t)(P.launches.size() - 1));
            P.iter_gates.push_back(P.total_gates);
            next_iter++;
        }
    }
    return true;
}

// Garbled-table ring (co-located solver): launch i owns the byte range [off[i], off[i] + len[i]) of a
// ring of `ring_bytes`, allocated in launch order with wrap-around; before the garbler overwrites the
// range it waits for the evaluation of wait[i], the newest earlier launch whose range overlaps (the
// evaluator runs in launch order, so every older overlapping launch is done by then too; -1: none).
// ring_bytes == 0 picks the largest launch plus room for what runs ahead of its evaluation: as much again, at most
// kRingSlackBytes (the large launches are the MAC launches, which alternate garble / evaluate anyway; what does run ahead
// are the small launches between them -- merges, inner products, dividers, reveals: tens of MB per iteration).  Returns the
// ring size.
inline size_t plan_table_ring(const Program &P, size_t ring_bytes, std::vector<size_t> &off, std::vector<int64_t> &wait) {
    const size_t align = 4096, nl = P.launches.size();
    const size_t tbytes = (size_t)P.max_launch_steps * 2048;
    const size_t min_ring = (tbytes + align - 1) / align * align;
    if (ring_bytes == 0) ring_bytes = min_ring + (min_ring < ring_slack_bytes() ? min_ring : ring_slack_bytes()) + align;
    if (ring_bytes < min_ring) ring_bytes = min_ring;
    off.resize(nl);
    wait.assign(nl, -1);
    std::vector<size_t> len(nl);
    size_t head = 0;
    for (size_t i = 0; i < nl; i++) {
        len[i] = ((size_t)P.launches[i].steps * 2048 + align - 1) / align * align;
        if (head + len[i] > ring_bytes) head = 0;
        off[i] = head;
        head += len[i];
    }
    for (size_t i = 0; i < nl; i++) {
        // scanning back may stop once more than a full ring of newer ranges has been passed: anything
        // older was overwritten by launches that waited for evaluations newer than it
        size_t seen = 0;
        for (size_t j = i; j-- > 0 && seen <= ring_bytes;) {
            seen += len[j];
            if (len[i] && len[j] && off[j] < off[i] + len[i] && off[i] < off[j] + len[j]) {
                wait[i] = (int64_t)j;
                break;
            }
        }
        // the garbler chain is in order: what an earlier launch waited for holds for this one too
        if (i > 0 && wait[i - 1] > wait[i]) wait[i] = wait[i - 1];
    }
    return ring_bytes;
}
}  // namespace gc
