// gc_cpu.cpp -- CPU checker / baseline for the garbled word machine.
// TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg).  The product (linreg-mpc_amd/) never links this.
//
// It runs the very same records (gc_exec.h) and circuits (gc_circuits.h) as
// the GPU kernels, but on host cores:
//   * gcc_plain_run   : plaintext bits, no crypto -- checks the lowering
//                       (gc_program.h) against the semantic oracle
//   * gcc_garble_run / gcc_eval_run : half-gates with AES-NI, gate for gate the
//                       protocol of gc_device.h (same hash, tweaks, table
//                       layout), so tables and labels are comparable bit for
//                       bit with the GPU's; this is also the "port" CPU
//                       baseline: one thread per role processes gates in
//                       program order, the execution structure of the
//                       reference's Obliv-C runtime.
// The reference's own runtime (Obliv-C + absentminded-crypto-kit) cannot be
// built here: gate-level parity with it is "unpinned".
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <wmmintrin.h>

#include <vector>

#include "../linreg-mpc_amd/csrc/gc_aes.h"
#include "../linreg-mpc_amd/csrc/gc_exec.h"

using namespace gc;

static AesTables g_t;
static __m128i g_rk[11];
static bool g_init = false;
static void init() {
    if (g_init) return;
    aes_build_tables(g_t, kFixedKey);
    for (int i = 0; i < 11; i++) g_rk[i] = _mm_loadu_si128((const __m128i *)&g_t.rk[4 * i]);
    g_init = true;
}

template <int N>
static inline void aesni_n(__m128i s[N]) {
    for (int b = 0; b < N; b++) s[b] = _mm_xor_si128(s[b], g_rk[0]);
    for (int r = 1; r < 10; r++)
        for (int b = 0; b < N; b++) s[b] = _mm_aesenc_si128(s[b], g_rk[r]);
    for (int b = 0; b < N; b++) s[b] = _mm_aesenclast_si128(s[b], g_rk[10]);
}
// the permutation of the gate hash (gc_aes.h): fixed-key AES, here through AES-NI
template <int N>
static inline void gate_perm_n(__m128i s[N]) { aesni_n<N>(s); }
// sigma(x) ^ tweak  (gc_aes.h: hash_prep)
static inline __m128i hprep(__m128i x, uint64_t tw) {
    __m128i sw = _mm_shuffle_epi32(x, 0x4E);                          // (x2,x3,x0,x1)
    __m128i hi = _mm_and_si128(x, _mm_set_epi32(-1, -1, 0, 0));       // (0,0,x2,x3)
    __m128i s = _mm_xor_si128(sw, hi);                                // (x2,x3,x0^x2,x1^x3)
    return _mm_xor_si128(s, _mm_set_epi64x(0, (long long)tw));
}

struct W64 {
    __m128i l[64];
};

template <bool GARBLER>
struct CpuBackend {
    typedef W64 W;
    static const bool kPairSteps = false;
    __m128i R;
    __m128i *words;
    __m128i *tab;
    uint64_t *decode;
    uint64_t step, launch_step0;
    uint64_t gates;

    W zero() const { W r; for (int i = 0; i < 64; i++) r.l[i] = _mm_setzero_si128(); return r; }
    W konst(uint64_t bits) const {
        W r;
        for (int i = 0; i < 64; i++) r.l[i] = (GARBLER && ((bits >> i) & 1)) ? R : _mm_setzero_si128();
        return r;
    }
    W XOR(const W &a, const W &b) const { W r; for (int i = 0; i < 64; i++) r.l[i] = _mm_xor_si128(a.l[i], b.l[i]); return r; }
    W NOTm(const W &a, uint64_t m) const {
        if (!GARBLER) return a;
        W r;
        for (int i = 0; i < 64; i++) r.l[i] = ((m >> i) & 1) ? _mm_xor_si128(a.l[i], R) : a.l[i];
        return r;
    }
    W sel(uint64_t m, const W &a, const W &b) const { W r; for (int i = 0; i < 64; i++) r.l[i] = ((m >> i) & 1) ? a.l[i] : b.l[i]; return r; }
    W bcast(const W &a, int src) const { W r; for (int i = 0; i < 64; i++) r.l[i] = a.l[src]; return r; }
    W bcast2(const W &a, int r) const { W o; for (int i = 0; i < 64; i++) o.l[i] = a.l[(i < 32 ? 0 : 32) + r]; return o; }
    W bblk(const W &a, int k) const { W o; const int h = 1 << k; for (int i = 0; i < 64; i++) o.l[i] = a.l[(i & ~(2 * h - 1)) | (h - 1)]; return o; }
    W shl(const W &a, int k) const { W r; for (int i = 0; i < 64; i++) r.l[i] = (i - k >= 0 && k < 64) ? a.l[i - k] : _mm_setzero_si128(); return r; }
    W shr(const W &a, int k) const { W r; for (int i = 0; i < 64; i++) r.l[i] = (i + k < 64 && k < 64) ? a.l[i + k] : _mm_setzero_si128(); return r; }
    W AND(const W &a, const W &b, uint64_t act) {
        W c;
        __m128i *slot = tab + (step - launch_step0) * 128;
        for (int lane = 0; lane < 64; lane++) {
            if (!((act >> lane) & 1)) { c.l[lane] = _mm_setzero_si128(); continue; }
            const uint64_t gid = step * 64 + (uint64_t)lane;
            gates++;
            if (GARBLER) {
                __m128i a0 = a.l[lane], b0 = b.l[lane];
                __m128i k[4] = {hprep(a0, 2 * gid), hprep(_mm_xor_si128(a0, R), 2 * gid),
                                hprep(b0, 2 * gid + 1), hprep(_mm_xor_si128(b0, R), 2 * gid + 1)};
                __m128i h[4] = {k[0], k[1], k[2], k[3]};
                gate_perm_n<4>(h);
                for (int i = 0; i < 4; i++) h[i] = _mm_xor_si128(h[i], k[i]);
                const int pa = _mm_cvtsi128_si32(a0) & 1, pb = _mm_cvtsi128_si32(b0) & 1;
                __m128i TG = _mm_xor_si128(_mm_xor_si128(h[0], h[1]), pb ? R : _mm_setzero_si128());
                __m128i WG = _mm_xor_si128(h[0], pa ? TG : _mm_setzero_si128());
                __m128i TE = _mm_xor_si128(_mm_xor_si128(h[2], h[3]), a0);
                __m128i WE = _mm_xor_si128(h[2], pb ? _mm_xor_si128(TE, a0) : _mm_setzero_si128());
                _mm_storeu_si128(slot + lane, TG);
                _mm_storeu_si128(slot + 64 + lane, TE);
                c.l[lane] = _mm_xor_si128(WG, WE);
            } else {
                __m128i av = a.l[lane], bv = b.l[lane];
                __m128i TG = _mm_loadu_si128(slot + lane), TE = _mm_loadu_si128(slot + 64 + lane);
                __m128i k[2] = {hprep(av, 2 * gid), hprep(bv, 2 * gid + 1)};
                __m128i h[2] = {k[0], k[1]};
                gate_perm_n<2>(h);
                h[0] = _mm_xor_si128(h[0], k[0]);
                h[1] = _mm_xor_si128(h[1], k[1]);
                const int sa = _mm_cvtsi128_si32(av) & 1, sb = _mm_cvtsi128_si32(bv) & 1;
                __m128i WG = _mm_xor_si128(h[0], sa ? TG : _mm_setzero_si128());
                __m128i WE = _mm_xor_si128(h[1], sb ? _mm_xor_si128(TE, av) : _mm_setzero_si128());
                c.l[lane] = _mm_xor_si128(WG, WE);
            }
        }
        step++;
        return c;
    }
    void AND2(const W &a1, const W &b1, uint64_t act1, const W &a2, const W &b2, uint64_t act2, W &c1, W &c2) {
        c1 = AND(a1, b1, act1);
        c2 = AND(a2, b2, act2);
    }
    W load(uint32_t id) const { W r; for (int i = 0; i < 64; i++) r.l[i] = _mm_loadu_si128(words + (size_t)id * 64 + i); return r; }
    W load2(uint32_t lo, uint32_t hi) const { W r; for (int i = 0; i < 64; i++) r.l[i] = _mm_loadu_si128(words + (size_t)(i < 32 ? lo : hi) * 64 + (i & 31)); return r; }
    W load2s(uint32_t lo, uint32_t hi, bool upper) const { return upper ? load2h(lo, hi) : load2(lo, hi); }
    W load2h(uint32_t lo, uint32_t hi) const { W r; for (int i = 0; i < 64; i++) r.l[i] = _mm_loadu_si128(words + (size_t)(i < 32 ? lo : hi) * 64 + 32 + (i & 31)); return r; }
    void store2(uint32_t lo, uint32_t hi, const W &v) { for (int i = 0; i < 64; i++) _mm_storeu_si128(words + (size_t)(i < 32 ? lo : hi) * 64 + (i & 31), v.l[i]); }
    void store(uint32_t id, const W &v) { for (int i = 0; i < 64; i++) _mm_storeu_si128(words + (size_t)id * 64 + i, v.l[i]); }
    void reveal(uint32_t slot, const W &v) {
        uint64_t m = 0;
        for (int i = 0; i < 64; i++) m |= (uint64_t)(_mm_cvtsi128_si32(v.l[i]) & 1) << i;
        if (decode) decode[slot] = m;
    }
};

extern "C" {

// ---- plaintext execution of records (checks the program lowering)
int gcc_plain_run(const Rec *recs, size_t nrec, int w, int p, uint64_t *words, uint64_t *decode,
                  uint64_t *steps, uint64_t *gates) {
    PlainMachine m(words, decode);
    for (size_t i = 0; i < nrec; i++) {
        if (recs[i].step0 != m.steps) return 1;   // the builder's step accounting must match execution
        exec_record(m, recs[i], w, p);
    }
    if (steps) *steps = m.steps;
    if (gates) *gates = m.gates;
    return 0;
}

// gate steps / AND gates of one record (the builder's cost model: rec_cost)
void gcc_rec_cost(uint32_t op, uint32_t cnt, int w, int p, uint64_t *steps, uint64_t *gates) {
    Rec r;
    r.op = op; r.cnt = cnt; r.dst = 0; r.a = 0; r.b = 0; r.c = 0; r.sa = 1; r.sb = 1; r.step0 = 0;
    rec_cost(r, w, p, *steps, *gates, 0);
}

// the same with the step order of the latency-bound GPU kernels (B::kPairSteps: independent gate steps
// of the multiplier issued as dual steps); step0 is not checked record by record, only the totals
struct PlainMachinePaired : PlainMachine {
    static const bool kPairSteps = true;
    PlainMachinePaired(uint64_t *w_, uint64_t *d_) : PlainMachine(w_, d_) {}
};
int gcc_plain_run_paired(const Rec *recs, size_t nrec, int w, int p, uint64_t *words, uint64_t *decode,
                         uint64_t *steps, uint64_t *gates) {
    PlainMachinePaired m(words, decode);
    for (size_t i = 0; i < nrec; i++) {
        if (recs[i].step0 != m.steps) return 1;
        exec_record(m, recs[i], w, p);
    }
    if (steps) *steps = m.steps;
    if (gates) *gates = m.gates;
    return 0;
}

// one word operation on many operand pairs (op-level fuzzing of gc_circuits.h against the semantic
// oracle): out[i] = op(a[i], b[i]); c is the public constant of OP_IDIVC.  Returns the gate steps of one op.
uint64_t gcc_plain_op(uint32_t op, int w, int p, uint32_t c, int paired, const uint64_t *a, const uint64_t *b,
                      uint64_t *out, size_t n) {
    uint64_t words[4];
    uint64_t steps = 0;
    for (size_t i = 0; i < n; i++) {
        words[0] = 0; words[1] = a[i]; words[2] = b[i]; words[3] = 0;
        Rec r;
        r.op = op; r.cnt = 1; r.dst = 3; r.a = 1; r.b = 2; r.c = c; r.sa = 1; r.sb = 1; r.step0 = 0;
        if (op == OP_IDIVC) r = idivc_rec(3, 1, c, w);
        if (paired) { PlainMachinePaired m(words, 0); exec_record(m, r, w, p); steps = m.steps; }
        else { PlainMachine m(words, 0); exec_record(m, r, w, p); steps = m.steps; }
        out[i] = words[3];
    }
    return steps;
}

// AES-128 under the seed's own key schedule (mirror of seed_keys / derive_R / gc_input_kernel in gc_engine.hip)
static void seed_rk(const uint8_t seed[16], __m128i rk[11]) {
    AesTables t;
    aes_build_tables(t, seed);
    for (int i = 0; i < 11; i++) rk[i] = _mm_loadu_si128((const __m128i *)&t.rk[4 * i]);
}
static inline __m128i aes_with(const __m128i rk[11], __m128i s) {
    s = _mm_xor_si128(s, rk[0]);
    for (int r = 1; r < 10; r++) s = _mm_aesenc_si128(s, rk[r]);
    return _mm_aesenclast_si128(s, rk[10]);
}
void gcc_derive_R(const uint8_t seed[16], uint8_t out[16]) {
    init();
    __m128i rk[11];
    seed_rk(seed, rk);
    __m128i h = aes_with(rk, _mm_set_epi32(2, 0, 0, 0));
    h = _mm_or_si128(h, _mm_set_epi32(0, 0, 0, 1));
    _mm_storeu_si128((__m128i *)out, h);
}

// mirror of gc_input_kernel (gc_engine.hip)
void gcc_input_labels(const uint8_t seed[16], const uint8_t R[16], const uint64_t *vals, uint32_t base, uint32_t n,
                      int w, uint8_t *wordsG, uint8_t *wordsE) {
    init();
    __m128i rk[11];
    seed_rk(seed, rk);
    __m128i r = _mm_loadu_si128((const __m128i *)R);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t id = base + k;
        for (int lane = 0; lane < 64; lane++) {
            uint64_t idx = (uint64_t)id * 64 + (uint64_t)lane;
            __m128i z = aes_with(rk, _mm_set_epi32(1, 0, (int)(uint32_t)(idx >> 32), (int)(uint32_t)idx));
            int bit = (int)((vals[k] >> lane) & 1);
            if (lane >= w) { z = _mm_setzero_si128(); bit = 0; }
            _mm_storeu_si128((__m128i *)wordsG + (size_t)id * 64 + lane, z);
            if (wordsE) _mm_storeu_si128((__m128i *)wordsE + (size_t)id * 64 + lane, bit ? _mm_xor_si128(z, r) : z);
        }
    }
}

// run records [0, nrec) of one or more launches on one thread
uint64_t gcc_garble_run(const Rec *recs, size_t nrec, int w, int p, const uint8_t R[16], uint8_t *words,
                        uint8_t *tab, uint64_t *decode, uint64_t launch_step0) {
    init();
    CpuBackend<true> be;
    be.R = _mm_loadu_si128((const __m128i *)R);
    be.words = (__m128i *)words;
    be.tab = (__m128i *)tab;
    be.decode = decode;
    be.launch_step0 = launch_step0;
    be.gates = 0;
    for (size_t i = 0; i < nrec; i++) {
        be.step = recs[i].step0;
        exec_record(be, recs[i], w, p);
    }
    return be.gates;
}
uint64_t gcc_eval_run(const Rec *recs, size_t nrec, int w, int p, uint8_t *words, uint8_t *tab, uint64_t *decode,
                      uint64_t launch_step0) {
    init();
    CpuBackend<false> be;
    be.R = _mm_setzero_si128();
    be.words = (__m128i *)words;
    be.tab = (__m128i *)tab;
    be.decode = decode;
    be.launch_step0 = launch_step0;
    be.gates = 0;
    for (size_t i = 0; i < nrec; i++) {
        be.step = recs[i].step0;
        exec_record(be, recs[i], w, p);
    }
    return be.gates;
}

void gcc_aes_encrypt(const uint8_t *in, uint8_t *out, size_t n) {
    init();
    for (size_t i = 0; i < n; i++) {
        __m128i s[1] = {_mm_loadu_si128((const __m128i *)in + i)};
        aesni_n<1>(s);
        _mm_storeu_si128((__m128i *)out + i, s[0]);
    }
}
// AES-128-CTR keystream with an arbitrary key: block c = AES_key(c), little-endian counter
// (mirror of ti_prg_kernel in linreg-mpc_amd/csrc/phase1.hip)
void gcc_aes_ctr(const uint8_t key[16], uint64_t first_block, uint64_t nblocks, uint8_t *out) {
    AesTables t;
    aes_build_tables(t, key);
    __m128i rk[11];
    for (int i = 0; i < 11; i++) rk[i] = _mm_loadu_si128((const __m128i *)&t.rk[4 * i]);
    for (uint64_t b = 0; b < nblocks; b++) {
        __m128i s = _mm_set_epi64x(0, (long long)(first_block + b));
        s = _mm_xor_si128(s, rk[0]);
        for (int r = 1; r < 10; r++) s = _mm_aesenc_si128(s, rk[r]);
        s = _mm_aesenclast_si128(s, rk[10]);
        _mm_storeu_si128((__m128i *)out + b, s);
    }
}
// ---- CPU mirror of the IKNP extension of linreg-mpc_amd/csrc/ot.hip (both roles in one call).
// seeds0/seeds1: 128 x 16 bytes; delta: 16 bytes; cbits: m bits packed LSB-first.
// Outputs: u (128 columns x m128 blocks), rows_t / rows_q (m128*128 x 16 bytes).
static inline __m128i hash1(__m128i x, uint64_t tw) {
    __m128i k = hprep(x, tw);
    __m128i h[1] = {k};
    aesni_n<1>(h);
    return _mm_xor_si128(h[0], k);
}
void gcc_iknp_extend(const uint8_t *seeds0, const uint8_t *seeds1, const uint8_t delta[16], const uint8_t *cbits,
                     uint64_t m, uint64_t ctr0, uint8_t *u, uint8_t *rows_t, uint8_t *rows_q) {
    init();
    const uint64_t m128 = (m + 127) / 128;
    std::vector<uint8_t> T0(128 * m128 * 16), Q(128 * m128 * 16), g1(m128 * 16);
    for (int j = 0; j < 128; j++) {
        gcc_aes_ctr(seeds0 + 16 * j, ctr0, m128, &T0[(size_t)j * m128 * 16]);
        gcc_aes_ctr(seeds1 + 16 * j, ctr0, m128, g1.data());
        const bool dj = (delta[j >> 3] >> (j & 7)) & 1;
        for (uint64_t k = 0; k < m128 * 16; k++) {
            uint8_t uu = (uint8_t)(T0[(size_t)j * m128 * 16 + k] ^ g1[k] ^ cbits[k]);
            u[(size_t)j * m128 * 16 + k] = uu;
            // sender: G(k^{delta_j}) ^ delta_j * u
            Q[(size_t)j * m128 * 16 + k] = dj ? (uint8_t)(g1[k] ^ uu) : T0[(size_t)j * m128 * 16 + k];
        }
    }
    memset(rows_t, 0, m128 * 128 * 16);
    memset(rows_q, 0, m128 * 128 * 16);
    for (uint64_t i = 0; i < m128 * 128; i++)
        for (int j = 0; j < 128; j++) {
            size_t byte = (size_t)j * m128 * 16 + (i >> 3);
            if ((T0[byte] >> (i & 7)) & 1) rows_t[i * 16 + (j >> 3)] |= (uint8_t)(1u << (j & 7));
            if ((Q[byte] >> (i & 7)) & 1) rows_q[i * 16 + (j >> 3)] |= (uint8_t)(1u << (j & 7));
        }
}
// Gilboa payload arithmetic on given rows: y, sender shares, receiver shares
void gcc_iknp_gilboa(const uint8_t *rows_t, const uint8_t *rows_q, const uint8_t delta[16], const uint64_t *a,
                     const uint64_t *b, uint64_t npairs, uint64_t n, int w, uint64_t tweak0, uint64_t *y,
                     uint64_t *share_s, uint64_t *share_r) {
    init();
    const uint64_t mask = w == 32 ? 0xffffffffull : ~0ull;
    __m128i D = _mm_loadu_si128((const __m128i *)delta);
    for (uint64_t q = 0; q < npairs; q++) {
        uint64_t ss = 0, sr = 0;
        for (uint64_t k = 0; k < n; k++)
            for (int bit = 0; bit < w; bit++) {
                uint64_t i = (q * n + k) * (uint64_t)w + (uint64_t)bit;
                __m128i qi = _mm_loadu_si128((const __m128i *)rows_q + i), ti = _mm_loadu_si128((const __m128i *)rows_t + i);
                uint64_t x0 = (uint64_t)_mm_cvtsi128_si64(hash1(qi, tweak0 + i)) & mask;
                uint64_t h1 = (uint64_t)_mm_cvtsi128_si64(hash1(_mm_xor_si128(qi, D), tweak0 + i)) & mask;
                uint64_t d = (b[q * n + k] << bit) & mask;
                y[i] = (x0 + d - h1) & mask;
                ss -= x0;
                uint64_t v = (uint64_t)_mm_cvtsi128_si64(hash1(ti, tweak0 + i)) & mask;
                if ((a[q * n + k] >> bit) & 1) v += y[i];
                sr += v;
            }
        share_s[q] = ss & mask;
        share_r[q] = sr & mask;
    }
}
void gcc_iknp_labels(const uint8_t *rows_t, const uint8_t *rows_q, const uint8_t delta[16], const uint8_t *choice,
                     const uint8_t *m0, const uint8_t *m1, uint64_t m, uint64_t tweak0, uint8_t *e, uint8_t *out) {
    init();
    __m128i D = _mm_loadu_si128((const __m128i *)delta);
    for (uint64_t i = 0; i < m; i++) {
        __m128i qi = _mm_loadu_si128((const __m128i *)rows_q + i), ti = _mm_loadu_si128((const __m128i *)rows_t + i);
        __m128i e0 = _mm_xor_si128(_mm_loadu_si128((const __m128i *)m0 + i), hash1(qi, tweak0 + i));
        __m128i e1 = _mm_xor_si128(_mm_loadu_si128((const __m128i *)m1 + i), hash1(_mm_xor_si128(qi, D), tweak0 + i));
        _mm_storeu_si128((__m128i *)e + 2 * i, e0);
        _mm_storeu_si128((__m128i *)e + 2 * i + 1, e1);
        _mm_storeu_si128((__m128i *)out + i, _mm_xor_si128(choice[i] ? e1 : e0, hash1(ti, tweak0 + i)));
    }
}
// portable T-table path of gc_aes.h on the host (the algorithm the GPU runs)
void gcc_aes_encrypt_ttable(const uint8_t *in, uint8_t *out, size_t n) {
    init();
    HostTab ht = {g_t.te0};
    for (size_t i = 0; i < n; i++) {
        uint32_t s[1][4];
        memcpy(s[0], in + 16 * i, 16);
        aes_encrypt_n<1, HostTab>(ht, g_t.rk, s);
        memcpy(out + 16 * i, s[0], 16);
    }
}
// H(x, t) through the shared scalar code of gc_aes.h (hash_n)
void gcc_gate_hash(const uint8_t *x, const uint64_t *tweak, uint8_t *out, size_t n) {
    init();
    for (size_t i = 0; i < n; i++) {
        Lbl l, o;
        memcpy(&l, x + 16 * i, 16);
        HostTab ht = {g_t.te0};
        hash_n<1, HostTab>(ht, g_t.rk, &l, &tweak[i], &o);
        memcpy(out + 16 * i, &o, 16);
    }
}
void gcc_hash(const uint8_t x[16], uint64_t tweak, uint8_t out[16]) {
    init();
    Lbl l;
    memcpy(&l, x, 16);
    HostTab ht = {g_t.te0};
    Lbl o;
    hash_n<1, HostTab>(ht, g_t.rk, &l, &tweak, &o);
    memcpy(out, &o, 16);
}

// ---- CPU baseline: garbler thread + evaluator thread over `nprod` products of
// the dominant unit (one OP_MAC record of `chunk` products each), evaluator
// following the garbler record by record (in-process hand-off of the table
// chunk).  Returns AND gates per second (garble + evaluate, wall clock).
struct BaseCtx {
    std::vector<Rec> recs;
    int w, p;
    uint8_t R[16];
    uint8_t *wordsG, *wordsE, *tab;
    volatile size_t garbled;   // records finished by the garbler
    volatile size_t evaluated; // records finished by the evaluator
    uint64_t rec_steps;        // gate steps per record
    uint64_t gatesG, gatesE;
};
static const size_t kRing = 8;  // table hand-off ring: the garbler runs at most kRing records ahead
static void *base_garbler(void *v) {
    BaseCtx *c = (BaseCtx *)v;
    uint64_t g = 0;
    for (size_t i = 0; i < c->recs.size(); i++) {
        while (i >= c->evaluated + kRing) _mm_pause();
        g += gcc_garble_run(&c->recs[i], 1, c->w, c->p, c->R, c->wordsG,
                            c->tab + (i % kRing) * c->rec_steps * 2048, 0, c->recs[i].step0);
        __sync_synchronize();
        c->garbled = i + 1;
    }
    c->gatesG = g;
    return 0;
}
static void *base_evaluator(void *v) {
    BaseCtx *c = (BaseCtx *)v;
    uint64_t g = 0;
    for (size_t i = 0; i < c->recs.size(); i++) {
        while (c->garbled <= i) _mm_pause();
        __sync_synchronize();
        g += gcc_eval_run(&c->recs[i], 1, c->w, c->p, c->wordsE, c->tab + (i % kRing) * c->rec_steps * 2048, 0,
                          c->recs[i].step0);
        __sync_synchronize();
        c->evaluated = i + 1;
    }
    c->gatesE = g;
    return 0;
}
// cpu_g / cpu_e: the logical CPUs the two threads are pinned to (-1: left to the scheduler).  Unpinned, the pair landed on
// SMT siblings of one core in some runs and on two cores in others: 5.2e7 .. 8.3e7 AND-gates/s on the same CPU model
// (BENCH_r02 / r03); bench.py picks two distinct physical cores and reports them.
double gcc_baseline_mac_on(int w, int p, uint32_t nrec, uint32_t chunk, int cpu_g, int cpu_e, uint64_t *and_gates, double *seconds);
double gcc_baseline_mac(int w, int p, uint32_t nrec, uint32_t chunk, uint64_t *and_gates, double *seconds) {
    return gcc_baseline_mac_on(w, p, nrec, chunk, -1, -1, and_gates, seconds);
}
double gcc_baseline_mac_on(int w, int p, uint32_t nrec, uint32_t chunk, int cpu_g, int cpu_e, uint64_t *and_gates, double *seconds) {
    init();
    BaseCtx c;
    c.w = w; c.p = p;
    c.garbled = 0;
    c.evaluated = 0;
    for (int i = 0; i < 16; i++) c.R[i] = (uint8_t)(0x3d * (i + 7));
    c.R[0] |= 1;
    // word file: word 0 zero, a-words 1..chunk, b-words chunk+1..2chunk, outputs after
    uint32_t nwords = 1 + 2 * chunk + 2 * nrec;
    c.wordsG = (uint8_t *)aligned_alloc(64, (size_t)nwords * 1024);
    c.wordsE = (uint8_t *)aligned_alloc(64, (size_t)nwords * 1024);
    std::vector<uint64_t> vals(2 * chunk);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < vals.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; vals[i] = x; }
    memset(c.wordsG, 0, (size_t)nwords * 1024);
    memset(c.wordsE, 0, (size_t)nwords * 1024);
    uint8_t seed[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    gcc_input_labels(seed, c.R, vals.data(), 1, 2 * chunk, w, c.wordsG, c.wordsE);
    uint64_t steps = 0, gates = 0, step_cursor = 0;
    for (uint32_t i = 0; i < nrec; i++) {
        Rec r;
        r.op = OP_MAC; r.cnt = chunk; r.dst = 1 + 2 * chunk + 2 * i; r.a = 1; r.b = 1 + chunk; r.c = 0;
        r.sa = 1; r.sb = 1; r.step0 = step_cursor;
        if (i == 0) rec_cost(r, w, p, steps, gates, 0);
        step_cursor += steps;
        c.recs.push_back(r);
    }
    c.rec_steps = steps;
    c.tab = (uint8_t *)aligned_alloc(64, (size_t)steps * 2048 * kRing);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_t tg, te;
    pthread_attr_t ag, ae;
    pthread_attr_init(&ag); pthread_attr_init(&ae);
    if (cpu_g >= 0) { cpu_set_t st; CPU_ZERO(&st); CPU_SET(cpu_g, &st); pthread_attr_setaffinity_np(&ag, sizeof st, &st); }
    if (cpu_e >= 0) { cpu_set_t st; CPU_ZERO(&st); CPU_SET(cpu_e, &st); pthread_attr_setaffinity_np(&ae, sizeof st, &st); }
    pthread_create(&tg, &ag, base_garbler, &c);
    pthread_create(&te, &ae, base_evaluator, &c);
    pthread_attr_destroy(&ag); pthread_attr_destroy(&ae);
    pthread_join(tg, 0);
    pthread_join(te, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    // self-check: decoded accumulators agree between the roles up to R (colour bits)
    free(c.wordsG); free(c.wordsE); free(c.tab);
    if (and_gates) *and_gates = c.gatesG;
    if (seconds) *seconds = sec;
    return (double)c.gatesG / sec;
}

}  // extern "C"
