// asan_program.cpp -- the host-side program lowering (gc_program.h) and the record executor (gc_exec.h) under
// AddressSanitizer + UBSan (CPU build only; tests/test_program_cpu.py compiles and runs it).
//
// For every configuration the word file and the decode array are heap blocks of EXACTLY n_words / n_reveal entries, so
// a record that reads or writes one word past what the builder allocated is an ASan report, not silent corruption (the
// failure of round 3: partial sums of a factorisation written past the scratch of dots()).  The run also checks what
// the engine relies on: ranges_ok(), consecutive gate-step numbering, launch slices that tile the record list.
//
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all asan_program.cpp -o asan_program
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>

#include "../../linreg-mpc_amd/csrc/gc_program.h"

using namespace gc;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

static int check_structure(const Program &P, const char *what) {
    uint64_t step = P.launches.empty() ? 0 : P.launches[0].step0;
    uint32_t rec = 0;
    for (size_t li = 0; li < P.launches.size(); li++) {
        const Launch &L = P.launches[li];
        if (L.first_rec != rec) { printf("%s: launch %zu does not start where launch %zu ended\n", what, li, li - 1); return 1; }
        if (li == P.prefix_launches && P.replicas > 1) step = L.step0;      // a sweep block's circuits start at their own offset
        if (L.step0 != step) { printf("%s: launch %zu step0 %llu, expected %llu\n", what, li, (unsigned long long)L.step0, (unsigned long long)step); return 1; }
        if (L.steps > P.cap_steps) { printf("%s: launch %zu above the cap\n", what, li); return 1; }
        rec += L.nrec;
        step += L.steps;
    }
    if (rec != P.recs.size()) { printf("%s: launches cover %u of %zu records\n", what, rec, P.recs.size()); return 1; }
    return 0;
}

// run every record on a plaintext machine whose word file is exactly n_words long
static int run_plain(const Program &P, const char *what) {
    std::unique_ptr<uint64_t[]> words(new uint64_t[P.n_words]);
    std::unique_ptr<uint64_t[]> dec(new uint64_t[P.n_reveal ? P.n_reveal : 1]);
    memset(words.get(), 0, sizeof(uint64_t) * P.n_words);
    const uint64_t mask = P.w == 64 ? ~0ull : 0xffffffffull;
    const size_t nin = P.nshares * (P.T + P.d);
    for (size_t i = 0; i < nin; i++) words[P.in_base + i] = (rnd() >> 8) & mask;     // any values: the control flow is data-independent
    PlainMachine m(words.get(), dec.get());
    uint64_t steps = 0;
    for (size_t i = 0; i < P.recs.size(); i++) {
        const uint64_t before = m.steps;
        exec_record(m, P.recs[i], P.w, P.p);
        steps += m.steps - before;
    }
    if (steps != P.total_steps || m.gates != P.total_gates) {
        printf("%s: executed %llu steps / %llu gates, the builder counted %llu / %llu\n", what, (unsigned long long)steps,
               (unsigned long long)m.gates, (unsigned long long)P.total_steps, (unsigned long long)P.total_gates);
        return 1;
    }
    return 0;
}

static int one(int alg, size_t d, int w, int p, int iters, size_t nshares, int normalize, int trace, size_t sweep, uint64_t cap, int kara) {
    char what[160];
    snprintf(what, sizeof what, "alg=%d d=%zu w=%d p=%d iters=%d shares=%zu norm=%d trace=%d sweep=%zu cap=%llu kara=%d", alg, d, w, p, iters,
             nshares, normalize, trace, sweep, (unsigned long long)cap, kara);
    program_karatsuba() = kara;
    Program P0;
    if (cap) P0.cap_steps = cap;
    if (sweep > 1) { P0.merge_hint = sweep; if (!cap) P0.cap_steps = kSweepCapSteps; }
    build_program(P0, alg, d, w, p, iters, nshares, normalize, 0x1234567ull, 0, trace);
    int bad = 0;
    if (P0.overflow || !P0.ranges_ok()) { printf("%s: builder reports overflow / ranges\n", what); bad = 1; }
    bad |= check_structure(P0, what);
    if (sweep <= 1) {
        bad |= run_plain(P0, what);
    } else {
        std::vector<uint64_t> lam(sweep);
        for (size_t i = 0; i < sweep; i++) lam[i] = (i + 1) * 0x1000ull;
        // the whole sweep in one program, and the second half as its own block (a rank of a multi-GPU sweep)
        for (int half = 0; half < 2; half++) {
            Program P;
            const size_t first = half ? sweep / 2 : 0, count = half ? sweep - sweep / 2 : sweep;
            if (!replicate_program(P, P0, count, lam.data() + first, first)) { printf("%s: sweep block refused\n", what); bad = 1; }
            if (P.overflow || !P.ranges_ok()) { printf("%s: sweep block reports overflow / ranges\n", what); bad = 1; }
            bad |= check_structure(P, what);
            bad |= run_plain(P, what);
        }
    }
    printf("%s %s: %zu records, %zu launches, %u words\n", bad ? "FAIL" : "ok  ", what, P0.recs.size(), P0.launches.size(), P0.n_words);
    return bad;
}

int main(int argc, char **argv) {
    const bool full = argc > 1 && !strcmp(argv[1], "full");
    int bad = 0;
    // small shapes: every algorithm, both widths, both input paths, traces, odd dimensions
    const size_t ds[] = {1, 2, 3, 5, 8, 13};
    for (size_t d : ds)
        for (int alg = 0; alg < 3; alg++)
            for (int w = 32; w <= 64; w += 32)
                bad |= one(alg, d, w, w == 64 ? 56 : 28, alg == ALG_CGD ? 3 : 0, 2 + d % 2, (int)(d & 1), alg == ALG_CGD, 1, 0, 1);
    // chunked dot products: caps small enough that every launch-shaping path splits (the round-3 failure needed d = 250)
    for (int alg = 0; alg < 3; alg++)
        for (int w = 32; w <= 64; w += 32) {
            bad |= one(alg, 40, w, w == 64 ? 56 : 28, alg == ALG_CGD ? 2 : 0, 2, 1, 0, 1, 1ull << 16, 1);
            bad |= one(alg, 67, w, w == 64 ? 56 : 28, alg == ALG_CGD ? 2 : 0, 3, 0, 0, 1, 1ull << 19, 1);
        }
    // Karatsuba on and off where it applies
    bad |= one(ALG_CGD, 132, 64, 56, 1, 2, 0, 0, 1, 0, 1);
    bad |= one(ALG_CGD, 132, 64, 56, 1, 2, 0, 0, 1, 0, 0);
    // sweeps: whole and as the block of a rank
    bad |= one(ALG_CGD, 10, 64, 56, 2, 2, 1, 0, 5, 0, 1);
    bad |= one(ALG_CHOLESKY, 9, 32, 28, 0, 3, 1, 0, 4, 0, 1);
    bad |= one(ALG_LDLT, 7, 64, 56, 0, 2, 1, 0, 3, 0, 1);
    bad |= one(ALG_CGD, 40, 64, 56, 1, 2, 1, 0, 8, 1ull << 18, 1);
    if (full) {
        bad |= one(ALG_CHOLESKY, 250, 32, 28, 0, 2, 0, 0, 1, 0, 1);
        bad |= one(ALG_CHOLESKY, 200, 64, 56, 0, 2, 0, 0, 1, 0, 1);
        bad |= one(ALG_LDLT, 200, 64, 56, 0, 2, 0, 0, 1, 0, 1);
        bad |= one(ALG_CGD, 100, 64, 56, 1, 2, 1, 0, 16, 0, 1);
    }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
