// gc_split.h -- "column-split" execution of latency-bound records (dividers, square roots, max trees: launches
// of so few records that the chain of dependent gate levels of ONE record is the run time).
//
// In the 4-wave mode (gc_device.h, MODE_QUAD) a lane owns one gate and hashes whole AES blocks: 16 table lookups
// per round, ~41 instructions per round and wave, and a wave issues them one after the other -- 2200 cycles per hash
// on a CU that is otherwise idle (scripts/exp/lat2.hip).  Here the state of a block is spread over the four lanes of
// a quad (lane 4j + c holds column c of block j; ShiftRows is three quad_perm DPP moves), a wave issues 4 lookups per
// round, and one hash of the 64 gates of a step takes four waves.  A workgroup is 16 waves:
//
//   * waves 0..3 ("glue" waves) run the record's circuit, wave c holding COLUMN c (one dword) of every label, lane =
//     gate: free-XOR glue is one instruction instead of four, lane moves are one ds_bpermute / v_readlane;
//   * at a gate step they put the operand columns into LDS, and all 16 waves hash: wave 4q + r takes hash q (operand
//     q of the step: a1, b1, a2, b2) of gates 16r .. 16r + 15 in quad layout, writes its column of the result back,
//     and the glue waves pick the results up: two workgroup barriers per posted gate step;
//   * whole additions -- nearly all levels of a divider or square root -- are posted as ONE job and walked by the hash
//     waves themselves with one barrier per prefix level (split_sk_add).
//
// Measured (scripts/exp/lat4.hip): 1484 cycles per 2-hash level against 2379, 2142 per 4-hash level against 2915.
//
// Gate numbering, tweaks, table rows and (for the garbler) the critical-path scheme with its (a0, b0) stash are those
// of MODE_QUAD with CRIT: a launch garbled here can be evaluated by either kernel and the other way round.
#pragma once
#include "gc_device.h"

namespace gc {

template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

// lane 4j + c: x = column c of label j, twc = tweak word of this column (c = 0: low, c = 1: high, else 0).
// Returns column c of H(x_j, tweak_j) = pi(sigma(x) ^ t) ^ sigma(x) ^ t  (gc_aes.h: hash_prep / hash_n)
// `beside(rnd)`: work that does not depend on the hash, called once per round between the lookups and their use -- a lone
// wave waits there for the LDS anyway (the adder levels compute the next level's addresses in those gaps)
struct HashNoHook { __device__ __forceinline__ void operator()(int) const {} };
template <class F = HashNoHook>
__device__ __forceinline__ uint32_t hash_split(const LdsTab4 &lt, const uint32_t rkl[11], uint32_t x, uint32_t twc, int c, F beside = F()) {
    const uint32_t t = quad_perm<0x4E>(x);                 // lanes 0,1 <- columns 2,3 ; lanes 2,3 <- columns 0,1
    const uint32_t k = (c < 2) ? (t ^ twc) : (x ^ t);
    uint32_t s = k ^ rkl[0];
#pragma unroll
    for (int rnd = 1; rnd < 10; rnd++) {
        // (looking up with the own column and routing the results as DPP operands of the XORs saves the three moves but
        // measured no faster: scripts/exp/lat4.hip 2 118 against 2 142 cycles, the divider 2 % slower)
        const uint32_t s1 = quad_perm<0x39>(s), s2 = quad_perm<0x4E>(s), s3 = quad_perm<0x93>(s);   // columns c+1, c+2, c+3
        const uint32_t v0 = lt.lkt(0, s, 0), v1 = lt.lkt(1, s1, 1), v2 = lt.lkt(2, s2, 2), v3 = lt.lkt(3, s3, 3);
        beside(rnd);
        s = xor3(xor3(v0, v1, rkl[rnd]), v2, v3);
    }
    const uint32_t s1 = quad_perm<0x39>(s), s2 = quad_perm<0x4E>(s), s3 = quad_perm<0x93>(s);
    const uint32_t v0 = lt.lk(s, 0), v1 = lt.lk(s1, 1), v2 = lt.lk(s2, 2), v3 = lt.lk(s3, 3);
    s = xor3(last_lo(v1, v0), last_hi(v3, v2), rkl[10]);
    return s ^ k;
}

// LDS exchange areas (dwords).  A plane holds one column of the 64 labels of a word; planes are 72 dwords apart so
// that both access patterns are conflict-free: glue wave c, lane g touches plane c at g (linear); hash lane 4j + c of
// gate block r touches plane c at 16r + j (bank 8c + 16r + j mod 32: two lanes per bank, as any 64-lane access).
enum {
    kSplitPlane = 72,
    kSplitWord = 4 * kSplitPlane,          // one operand / one hash result: 4 planes
    kSplitOp = 0,                          // operands a1, b1, a2, b2 (adder: x, y, carry-in)
    kSplitX = 4 * kSplitWord,              // results of hash 0..3 (adder: sum, final generate word)
    kSplitKs = 8 * kSplitWord,             // adder levels: two buffers of hash results (level parity; words 0, 1 of each used)
    kSplitGs = 16 * kSplitWord,            // adder levels: two published states (G, P), alternating
    kSplitDesc = 20 * kSplitWord,          // kind | n << 8, step (2), act1 (2), act2 (2): two 16-byte stores (an addition reads the first)
    kSplitZero = 20 * kSplitWord + 8,      // a dword that stays zero: the "no word here" operand of the adder levels
    kSplitWords = 20 * kSplitWord + 16
};

struct SplitDesc {
    uint32_t kind;          // 0: record finished, 1: one gate step (hashes 0, 1), 2: two gate steps (hashes 0..3),
                            // 3: a whole addition over act1 = lanes(n) (split_sk_add)
    uint64_t act1, act2;    // active gates of the step(s)
    uint64_t step;          // global index of the (first) gate step
};

// what every wave of the workgroup needs for the hash phase
struct SplitHashCtx {
    LdsTab4 lt;
    uint32_t rkl[11];       // round-key column lane & 3
    uint32_t Rq;            // column lane & 3 of the garbler's offset R (0 for the evaluator)
    uint32_t *sx;           // LDS exchange areas
    __attribute__((address_space(3))) uint32_t *sxl;   // the same, as an LDS pointer (32-bit arithmetic in the adder levels)
    uint32_t *tabw;         // this launch's table buffer, as dwords
    uint64_t launch_step0;
    int wave, lane;
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }   // unsigned: no sign extension
__device__ __forceinline__ void st_u32_global(uint32_t *p, uint32_t v) { *(__attribute__((address_space(1))) uint32_t *)p = v; }
__device__ __forceinline__ uint32_t ld_u32_global(const uint32_t *p) { return *(const __attribute__((address_space(1))) uint32_t *)p; }
__device__ __forceinline__ void st_u32_lds(uint32_t *p, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *)p = v; }
__device__ __forceinline__ uint32_t ld_u32_lds(const uint32_t *p) { return *(const __attribute__((address_space(3))) uint32_t *)p; }

// One hash of one gate step: wave 4q + r hashes operand (q & 1) of gates 16r .. 16r + 15 of step `st`; a_op / b_op are
// column c of the step's two operands for this lane's gate.  The result column goes to xdst (one word: 4 planes).
//   garbler  : x_a = H(a0 ^ pa R, 2g) ^ (pa & pb) R,  x_b = H(b0 ^ pb R, 2g + 1);  c0 = x_a ^ x_b   (gc_device.h: CRIT);
//              rows 0 / 1 of the step receive a0 / b0 for gc_tabfill_kernel
//   evaluator: x_a = H(a, 2g) ^ sa TG,  x_b = H(b, 2g + 1) ^ sb (TE ^ a);  c = x_a ^ x_b
// All 64 lanes of the wave are active here (the quad_perm moves read neighbouring lanes).
__device__ __forceinline__ uint32_t *split_row(const SplitHashCtx &hc, int q, uint64_t st) {
    const int gate = 16 * (hc.wave & 3) + (hc.lane >> 2);
    return hc.tabw + ((st - hc.launch_step0) * 128 + (uint64_t)((q & 1) * 64 + gate)) * 4 + (hc.lane & 3);
}
// tg (evaluator): this lane's dword of the step's TG (q even) / TE (q odd) row, loaded by the caller
template <bool GARBLER>
__device__ __forceinline__ void split_hash_core(const SplitHashCtx &hc, int q, uint64_t act, uint64_t st, uint32_t a_op, uint32_t b_op,
                                                uint32_t *xdst, uint32_t tg) {
    const int r = hc.wave & 3, c = hc.lane & 3, gate = 16 * r + (hc.lane >> 2);
    uint32_t *row = split_row(hc, q, st);
    uint32_t *xo = xdst + c * kSplitPlane + gate;
    const uint32_t act16 = (uint32_t)(act >> (16 * r)) & 0xffffu;     // this block's gates (wave-uniform)
    if (act16 == 0) {
        st_u32_lds(xo, 0u);
        if (GARBLER) st_u32_global(row, 0u);
        return;
    }
    const bool on = (act16 >> (hc.lane >> 2)) & 1u;
    const uint32_t v = (q & 1) ? b_op : a_op;
    if (GARBLER) st_u32_global(row, on ? v : 0u);
    const uint32_t colour = quad_perm<0x00>(v) & 1u;                                   // lsb of column 0 of the own operand
    const uint32_t x = GARBLER ? (v ^ (colour ? hc.Rq : 0u)) : v;
    const uint32_t tlo = ((uint32_t)st << 7) | (uint32_t)(2 * gate + (q & 1)), thi = (uint32_t)(st >> 25);
    const uint32_t twc = (c == 0) ? tlo : (c == 1) ? thi : 0u;
    uint32_t h = hash_split(hc.lt, hc.rkl, x, twc, c);
    if (GARBLER) {
        if (!(q & 1)) h ^= (colour & quad_perm<0x00>(b_op) & 1u) ? hc.Rq : 0u;
    } else {
        h ^= colour ? ((q & 1) ? (tg ^ a_op) : tg) : 0u;
    }
    st_u32_lds(xo, on ? h : 0u);
}

// Hash phase of a posted gate step (kind 1 / 2), between the two barriers: operands come from the LDS operand area.
template <bool GARBLER>
__device__ __forceinline__ void split_hash_phase(const SplitHashCtx &hc, const SplitDesc &d) {
    const int q = hc.wave >> 2, r = hc.wave & 3;
    if (d.kind == 1 && q >= 2) return;
    const int c = hc.lane & 3, gate = 16 * r + (hc.lane >> 2);
    const uint32_t *op = hc.sx + kSplitOp + c * kSplitPlane + gate;
    const uint32_t a_op = ld_u32_lds(op + (q & ~1) * kSplitWord), b_op = ld_u32_lds(op + (q | 1) * kSplitWord);
    const uint64_t st = d.step + (uint64_t)(q >> 1);
    const uint32_t tg = GARBLER ? 0u : ld_u32_global(split_row(hc, q, st));            // in flight during the hash
    split_hash_core<GARBLER>(hc, q, (q < 2) ? d.act1 : d.act2, st, a_op, b_op, hc.sx + kSplitX + q * kSplitWord, tg);
}

// A whole addition (gc_circuits.h: Circ::add_generic, the Sklansky prefix adder) run by the hash waves in quad layout with
// ONE barrier per level.  Posted by the glue waves with x, y and the carry-in word in operand words 0..2.  Gate steps, their
// order, lanes and activity masks are exactly those of Circ::add_generic: the first AND, then one step per level k, in which
// lane i with bit k set holds the G-gate of node i (P_i & G_m, m = the top lane of the lower half of i's block of 2^(k+1))
// and lane i - 2^k the P-gate of node i (P_i & P_m).
//
// A level is a single gate step: two hashes per gate, waves q = 0 (operand a) and q = 1 (operand b), eight of the sixteen
// waves.  Wave q = 2 keeps the published state one level behind:
//   state after level k - 1 at lane t:  G = G_pub(k-2)[t] ^ (bit k-1 of t ? X0 ^ X1 of level k-1 at t : 0)
//                                       P = (t was a P-node of level k-1 ? X0 ^ X1 of level k-1 at t - 2^(k-1) : P_pub(k-2)[t])
// (X0, X1: the two hash results of a gate; their XOR is the gate's output label).  Every hash lane rebuilds what it needs
// from what the last barrier made visible; results and published states alternate between two buffers.
//
// Round 5 (profiles/r4_split_trace.txt: 430-1020 cycles between a level's barrier and its first AES round, 200-700 more at
// the next barrier waiting for the slower wave, 1 750-2 900 between two additions):
//   * a hash wave rebuilds ITS OWN operand first -- every operand is the XOR of three LDS words, words a lane does not need
//     read a zero dword -- and hashes as soon as those three are back.  The OTHER operand enters only after the hash (the
//     garbler's a-wave needs the colour of b, the evaluator's b-wave the label a for its second ciphertext; the garbler's
//     b-wave and the evaluator's a-wave need nothing): its loads are issued before the hash and picked up behind it;
//   * the addresses of level k + 1 do not depend on data: they are computed while level k's loads are in flight, so the
//     path from the barrier to the first AES round is three loads and two XORs;
//   * the roles q = 0 / 1 are compiled separately (no per-lane selects between the two operand sets);
//   * the sum and the carry-out are formed by the glue waves straight from the last level's results (they are visible to
//     every wave after that level's barrier): no final pass in quad layout, no second hand-over barrier per addition.
#ifndef GC_SPLIT_TRACE
#define GC_SPLIT_TRACE 0      /* timing experiments only (scripts/exp/split_trace.py): s_memtime stamps of the adder levels of record 0 */
#endif
#if GC_SPLIT_TRACE
static __device__ uint64_t g_split_trace[2 * 8192];
static __device__ uint32_t g_split_trace_n[2];
struct SkTrace { bool on; int w; uint32_t i; };
#define SPLIT_STAMP(tag)                                                                                         \
    if (tr.on) {                                                                                                 \
        uint64_t t_;                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                                         \
        if (hc.lane == 0 && tr.i < 8192) g_split_trace[tr.w * 8192 + tr.i] = (t_ << 4) | (uint64_t)(tag);        \
        tr.i++;                                                                                                  \
    }
// GC_SPLIT_TRACE == 2: stamps of the POSTED gate steps (AND / AND2: the multiplier's levels) instead of the adder levels
#define SPLIT_STAMP2(wv, tag)                                                                                    \
    if (GC_SPLIT_TRACE == 2 && blockIdx.x == 0) {                                                                \
        uint64_t t_;                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                                         \
        const uint32_t i_ = g_split_trace_n[wv];                                                                 \
        if ((threadIdx.x & 63) == 0) { if (i_ < 8192) g_split_trace[(wv) * 8192 + i_] = (t_ << 4) | (uint64_t)(tag); g_split_trace_n[wv] = i_ + 1; } \
    }
#else
struct SkTrace {};
#define SPLIT_STAMP(tag)
#define SPLIT_STAMP2(wv, tag)
#endif

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// the three LDS words whose XOR is an operand: own (hashed by this wave) and other (needed after the hash, or not at all)
struct SkPtrs {
    const lds_u32 *o0, *o1, *o2;
    const lds_u32 *p0, *p1, *p2;
};

// number of prefix levels of an n-lane addition
__device__ __forceinline__ int sk_levels(int n) { return n <= 1 ? 0 : 32 - __builtin_clz((uint32_t)(n - 1)); }

// Operand words of level k at gate g (k = -1: the first AND of the addition, a = x ^ cin, b = y ^ cin at the own gate):
//   a = P of the own node t (t = g for a G-gate lane, g + 2^k for a P-gate host): two words
//   b = G of m (G-gate lane) or P of m (host), m = top lane of the lower half of the block: three words
// in plane `pl`.  They depend on k, g and the plane only -- not on the operands, the width of the addition or the activity
// masks.  Level 0 reads the first AND's results (buffer 0) and the operand words; level k > 0 the results of level k - 1
// and the state published for level k - 2.
__device__ __forceinline__ void sk_words_a(const lds_u32 *sx, int k, int g, int pl, const lds_u32 *&a0, const lds_u32 *&a1) {
    const lds_u32 *opx = sx + kSplitOp + pl, *opy = opx + kSplitWord, *opc = opx + 2 * kSplitWord;
    if (k < 0) { a0 = opx + g; a1 = opc + g; return; }
    const int h = 1 << k, hp = h >> 1;
    const int t = ((g >> k) & 1) ? g : ((g + h) & 63);
    if (k == 0) { a0 = opx + t; a1 = opy + t; return; }
    const lds_u32 *X0 = sx + kSplitKs + ((k & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;           // results of level k - 1
    const lds_u32 *Pp = sx + kSplitGs + (((k - 1) & 1) * 2 + 1) * kSplitWord + pl;                          // published P, state k - 2
    const bool tp = (g >> (k - 1)) & 1;                 // t was a P-node of level k - 1 (t >= 2^k, and t = g mod 2^k)
    const int ti = tp ? t - hp : t;
    a0 = (tp ? X0 : Pp) + ti;
    a1 = tp ? X1 + ti : sx + kSplitZero;
}
__device__ __forceinline__ void sk_words_b(const lds_u32 *sx, int k, int g, int pl, const lds_u32 *&b0, const lds_u32 *&b1, const lds_u32 *&b2) {
    const lds_u32 *opx = sx + kSplitOp + pl, *opy = opx + kSplitWord, *opc = opx + 2 * kSplitWord;
    const lds_u32 *zero = sx + kSplitZero;
    if (k < 0) { b0 = opy + g; b1 = opc + g; b2 = zero; return; }
    const int h = 1 << k, hp = h >> 1;
    const bool up = (g >> k) & 1;
    const int mb = (g & ~(2 * h - 1)) | (h - 1);
    const lds_u32 *X0 = sx + kSplitKs + ((k & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;
    if (k == 0) {
        b0 = (up ? X0 : opx) + mb;
        b1 = (up ? X1 : opy) + mb;
        b2 = up ? opc + mb : zero;
        return;
    }
    const lds_u32 *Gp = sx + kSplitGs + (((k - 1) & 1) * 2) * kSplitWord + pl;                               // published G, state k - 2
    const int mi = up ? mb : ((mb - hp) & 63);          // (m is a G-node and, for hosts, a P-node of level k - 1)
    b0 = X0 + mi;
    b1 = X1 + mi;
    b2 = up ? Gp + mb : zero;
}
// own / other operand of hash role Q (0: a, 1: b): own in plane pl, other in plane plo
template <int Q>
__device__ __forceinline__ void sk_ptrs_own(SkPtrs &r, const lds_u32 *sx, int k, int g, int pl) {
    if (Q == 0) { sk_words_a(sx, k, g, pl, r.o0, r.o1); r.o2 = sx + kSplitZero; }
    else sk_words_b(sx, k, g, pl, r.o0, r.o1, r.o2);
}
template <int Q>
__device__ __forceinline__ void sk_ptrs_oth(SkPtrs &r, const lds_u32 *sx, int k, int g, int plo) {
    if (Q == 0) sk_words_b(sx, k, g, plo, r.p0, r.p1, r.p2);
    else { sk_words_a(sx, k, g, plo, r.p0, r.p1); r.p2 = sx + kSplitZero; }
}

// what a hash lane knows about a level before the level's operands exist: where they will be, whether its gate is active,
// its tweak column and its dword of the step's table row.  Computed one level ahead (sk_plan), off the dependent chain.
struct SkPlan {
    SkPtrs ptr;
    uint32_t twc;           // tweak word of this lane's column (c = 0: low, c = 1: high, else 0)
    bool on;                // this lane's gate is active in the step
    bool any;               // some gate of this wave's sixteen is (wave-uniform)
};
// activity of gate g in level k of an n-lane addition (Circ::add_generic: bitk | host): a G-gate lane (bit k of g set) is
// active below n; a lane with bit k clear hosts the P-gate of node g + 2^k if that node's P is still needed (g >= 2^(k+1))
// and the node is below n.  k = -1: the first AND.
__device__ __forceinline__ bool sk_gate_on(int k, int g, int n) {
    if (k < 0) return g < n;
    const int h = 1 << k;
    return ((g >> k) & 1) ? (g < n) : (g >= 2 * h && g + h < n);
}
// part 0 .. 3 of the plan of level k (one part per AES round of the level before: sk_finish)
template <bool GARBLER, int Q>
__device__ __forceinline__ void sk_plan_part(SkPlan &r, int part, const lds_u32 *sx, int k, int g, int c, int pl, int plo, int n, uint64_t st) {
    constexpr bool kNeedOth = GARBLER ? (Q == 0) : (Q == 1);
    if (part == 0) sk_ptrs_own<Q>(r.ptr, sx, k, g, pl);
    if (part == 1) { if (kNeedOth) sk_ptrs_oth<Q>(r.ptr, sx, k, g, plo); else { r.ptr.p0 = r.ptr.p1 = r.ptr.p2 = sx + kSplitZero; } }
    if (part == 2) {
        const uint32_t tlo = ((uint32_t)st << 7) | (uint32_t)(2 * g + Q), thi = (uint32_t)(st >> 25);
        r.twc = (c == 0) ? tlo : (c == 1) ? thi : 0u;
    }
    if (part == 3) {
        r.on = sk_gate_on(k, g, n);
        r.any = __builtin_amdgcn_ballot_w64(r.on) != 0ull;
    }
}
template <bool GARBLER, int Q>
__device__ __forceinline__ SkPlan sk_plan(const lds_u32 *sx, int k, int g, int c, int pl, int plo, int n, uint64_t st) {
    SkPlan r;
    for (int part = 0; part < 4; part++) sk_plan_part<GARBLER, Q>(r, part, sx, k, g, c, pl, plo, n, st);
    return r;
}

// One hash of one level by role Q (wave 4Q + r: gates 16r .. 16r + 15), in two parts: sk_load issues the LDS reads of the
// operand words, sk_finish hashes.
//   garbler  : x_a = H(a0 ^ pa R, 2g) ^ (pa & pb) R,  x_b = H(b0 ^ pb R, 2g + 1);  rows 0 / 1 of the step receive a0 / b0
//   evaluator: x_a = H(a, 2g) ^ sa TG,  x_b = H(b, 2g + 1) ^ sb (TE ^ a)
// The other operand is read from plane 0 by the garbler (only its colour matters) and from the own plane by the evaluator.
struct SkVals { uint32_t l0, l1, l2, m0, m1, m2; };
template <bool GARBLER, int Q>
__device__ __forceinline__ SkVals sk_load(const SkPtrs &ptr) {
    constexpr bool kNeedOth = GARBLER ? (Q == 0) : (Q == 1);
    SkVals v;
    // operand a is the XOR of two words, operand b of three (sk_words_a, sk_words_b): the third pointer of an a-set is never read
    v.l0 = *ptr.o0; v.l1 = *ptr.o1; v.l2 = (Q == 1) ? *ptr.o2 : 0u;
    v.m0 = v.m1 = v.m2 = 0u;
    if (kNeedOth) { v.m0 = *ptr.p0; v.m1 = *ptr.p1; v.m2 = (Q == 0) ? *ptr.p2 : 0u; }
    return v;
}
// `beside`: work that does not depend on the hash (the plan of the next level); called inside the hash's basic block so that
// the scheduler can put it into the waits of the AES rounds
template <bool GARBLER, int Q, class F>
__device__ __forceinline__ void sk_finish(const SplitHashCtx &hc, const SkPlan &lv, const SkVals &w, lds_u32 *xo, uint32_t *row, uint32_t tg, F beside) {
    const int c = hc.lane & 3;
    const bool on = lv.on;
    const uint32_t twc = lv.twc;
    if (!lv.any) {
        *xo = 0u;
        if (GARBLER) st_u32_global(row, 0u);
        for (int rnd = 1; rnd < 10; rnd++) beside(rnd);
        return;
    }
    const uint32_t v = xor3(w.l0, w.l1, w.l2);
    if (GARBLER) st_u32_global(row, on ? v : 0u);
    const uint32_t colour = quad_perm<0x00>(v) & 1u;                                   // lsb of column 0 of the own operand
    const uint32_t x = GARBLER ? (v ^ (colour ? hc.Rq : 0u)) : v;
    uint32_t h = hash_split(hc.lt, hc.rkl, x, twc, c, beside);
    if (GARBLER) {
        if (Q == 0) h ^= (colour & (w.m0 ^ w.m1 ^ w.m2) & 1u) ? hc.Rq : 0u;
    } else {
        h ^= colour ? ((Q == 1) ? (tg ^ xor3(w.m0, w.m1, w.m2)) : tg) : 0u;
    }
    *xo = on ? h : 0u;
}

// hash role Q of a whole addition: 1 + sk_levels(n) levels, one barrier behind each.  `pre`: the first AND's operand
// words, already on their way (the hash waves issue them together with the read of the job descriptor)
template <bool GARBLER, int Q>
__device__ __forceinline__ void split_sk_hash(const SplitHashCtx &hc, int n, uint64_t st, const SkVals *pre) {
    const int r = hc.wave & 3, c = hc.lane & 3, g = 16 * r + (hc.lane >> 2);
    SkTrace tr;
    (void)tr;
#if GC_SPLIT_TRACE
    tr.on = GC_SPLIT_TRACE == 1 && blockIdx.x == 0 && (hc.wave == 0 || hc.wave == 4);
    tr.w = hc.wave == 4 ? 1 : 0;
    tr.i = tr.on ? g_split_trace_n[tr.w] : 0u;
    SPLIT_STAMP(1)                                                                  /* addition entered (after the hand-over barrier) */
#endif
    lds_u32 *sx = hc.sxl;
    const int pl = c * kSplitPlane, plo = GARBLER ? 0 : pl;
    const int L = sk_levels(n);
    // evaluator: this lane's dword of a level's ciphertext row is fetched TWO levels ahead, into two registers in turn (every
    // step index of the addition is known): the first AND and the odd levels use tA, the even levels tB; a register is
    // reloaded right behind its use.  (Rounds 3-5a fetched one level ahead and copied `tg = tgn` at the head of the loop: a
    // wait for a load issued one level -- ~1 900 cycles -- ago, right behind the barrier; the evaluator's levels were 200-380
    // cycles longer than the garbler's, profiles/r5_split_trace.txt.)  The level loop is unrolled by two for that: no copies.
    uint32_t *row = split_row(hc, Q, st);
    uint32_t tA = 0, tB = 0;
    if (!GARBLER) {
        // (every load below is issued whether or not its level exists -- a level that does not exist reads the current row
        // again: with a load on one path only, the wait in front of the OTHER register's use has to be for everything)
        tA = ld_u32_global(row);
        tB = ld_u32_global(row + (L > 0 ? 512 : 0));
    }
    // first AND: G = ((x ^ cin) & (y ^ cin)) ^ cin.  Results of level k go to buffer (k + 1) & 1; this step counts as level -1
    SkPlan cur = sk_plan<GARBLER, Q>(sx, -1, g, c, pl, plo, n, st), nxt;
    SkVals w = pre ? *pre : sk_load<GARBLER, Q>(cur.ptr);
    sk_finish<GARBLER, Q>(hc, cur, w, sx + kSplitKs + Q * kSplitWord + pl + g, row, tA,
                          [&](int rnd) { if (rnd >= 2 && rnd < 6) sk_plan_part<GARBLER, Q>(nxt, rnd - 2, sx, 0, g, c, pl, plo, n, st + 1); });
    if (!GARBLER) tA = ld_u32_global(row + (L > 1 ? 2 * 512 : 0));                  // row of level 1
    st += 1;
    row += 512;
    SPLIT_STAMP(2)                                                                  /* first AND hashed, result stored */
    lds_barrier();
    SPLIT_STAMP(3)                                                                  /* past the barrier */
    auto level = [&](int k, uint32_t &tg) {
        cur = nxt;
        w = sk_load<GARBLER, Q>(cur.ptr);
        sk_finish<GARBLER, Q>(hc, cur, w, sx + kSplitKs + (((k + 1) & 1) * 4 + Q) * kSplitWord + pl + g, row, tg,
                              [&](int rnd) { if (rnd >= 2 && rnd < 6) sk_plan_part<GARBLER, Q>(nxt, rnd - 2, sx, k + 1, g, c, pl, plo, n, st + 1); });
        if (!GARBLER) tg = ld_u32_global(row + (k + 2 < L ? 2 * 512 : 0));          // row of level k + 2, into the register just used
        SPLIT_STAMP(5)                                                              /* hashed, result stored */
        st += 1;
        row += 512;
        lds_barrier();
        SPLIT_STAMP(3)
    };
    int k = 0;
    for (; k + 1 < L; k += 2) { level(k, tB); level(k + 1, tA); }
    if (k < L) level(k, tB);
#if GC_SPLIT_TRACE
    if (tr.on && hc.lane == 0) g_split_trace_n[tr.w] = tr.i < 8192 ? tr.i : 8192;
#endif
}

// the waves that do not hash in an addition: wave q = 2 publishes the state one level behind, wave q = 3 only keeps the barriers
__device__ __forceinline__ void split_sk_side(const SplitHashCtx &hc, int n) {
    const int q = hc.wave >> 2, r = hc.wave & 3, c = hc.lane & 3, g = 16 * r + (hc.lane >> 2);
    lds_u32 *sx = hc.sxl;
    const int pl = c * kSplitPlane;
    const lds_u32 *opx = sx + kSplitOp + pl, *opy = opx + kSplitWord, *opc = opx + 2 * kSplitWord;
    const int L = sk_levels(n);
    lds_barrier();                                                                  // the first AND
    for (int k = 0; k < L; k++) {
        if (q == 2) {
            // publish state k - 1 of the own lane (read at level k + 1; after the last level, by the glue waves)
            const int h = 1 << k, hp = h >> 1;
            const lds_u32 *X0 = sx + kSplitKs + ((k & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;      // results of level k - 1
            const lds_u32 *Gp = sx + kSplitGs + (((k - 1) & 1) * 2) * kSplitWord + pl, *Pp = Gp + kSplitWord;  // published state k - 2
            lds_u32 *Go = sx + kSplitGs + ((k & 1) * 2) * kSplitWord + pl;
            const uint32_t x0 = X0[g], x1 = X1[g];
            uint32_t Gs, Ps;
            if (k == 0) {
                Gs = x0 ^ x1 ^ opc[g];
                Ps = opx[g] ^ opy[g];
            } else {
                const bool gp = ((g >> (k - 1)) & 1) && g >= h;
                const int gi = gp ? g - hp : g;
                const uint32_t p1 = (gp ? X0 : Pp)[gi], p2 = X1[gi], go = Gp[g];
                Gs = go ^ (((g >> (k - 1)) & 1) ? (x0 ^ x1) : 0u);
                Ps = gp ? (p1 ^ p2) : p1;
            }
            Go[g] = Gs;
            Go[kSplitWord + g] = Ps;
        }
        lds_barrier();
    }
}

// The circuit backend of the glue waves: W is ONE column of a word's labels (wave = column, lane = gate).
template <bool GARBLER>
struct SplitBackend {
    typedef uint32_t W;
    static const bool kPairSteps = true;       // same step numbering as MODE_QUAD
    SplitHashCtx hc;
    uint32_t Rc;                               // column `wave` of R
    Lbl *words;
    uint64_t *decode;
    uint64_t step;
    int wave, lane;

    __device__ __forceinline__ W zero() const { return 0u; }
    __device__ __forceinline__ bool bit(uint64_t m) const { return __builtin_amdgcn_inverse_ballot_w64(m); }
    __device__ __forceinline__ W rmask(uint64_t m) const { return bit(m) ? Rc : 0u; }
    __device__ __forceinline__ W konst(uint64_t bits) const { return GARBLER ? rmask(bits) : 0u; }
    __device__ __forceinline__ W XOR(W a, W b) const { return a ^ b; }
    __device__ __forceinline__ W NOTm(W a, uint64_t m) const { return GARBLER ? (a ^ rmask(m)) : a; }
    __device__ __forceinline__ W sel(uint64_t m, W a, W b) const { return bit(m) ? a : b; }
    __device__ __forceinline__ W bcast(W a, int src) const { return (uint32_t)__builtin_amdgcn_readlane((int)a, src); }
    __device__ __forceinline__ W bcast2(W a, int r) const { return sel(0xffffffffull, bcast(a, r), bcast(a, 32 + r)); }
    __device__ __forceinline__ W pull(W a, int from, bool ok) const {
        uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((from & 63) << 2, (int)a);
        return ok ? v : 0u;
    }
    __device__ __forceinline__ W bblk(W a, int k) const { const int h = 1 << k; return pull(a, (lane & ~(2 * h - 1)) | (h - 1), true); }
    // a shift by one lane is a DPP move over the whole wave (wave_shr:1 / wave_shl:1, zero into the end lane): the remainder
    // of a divider moves up one lane per quotient bit, a square root's by two -- a ds_bpermute there is an LDS round trip on
    // the glue waves' dependent chain
    __device__ __forceinline__ W up1(W a) const { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xf, 0xf, true); }
    __device__ __forceinline__ W down1(W a) const { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x130, 0xf, 0xf, true); }
    __device__ __forceinline__ W shl(W a, int k) const {
        if (k == 1) return up1(a);
        if (k == 2) return up1(up1(a));
        return k >= 64 ? 0u : pull(a, lane - k, lane >= k);
    }
    __device__ __forceinline__ W shr(W a, int k) const {
        if (k == 1) return down1(a);
        return k >= 64 ? 0u : pull(a, lane + k, lane + k < 64);
    }

    __device__ __forceinline__ const uint32_t *col(const Lbl *p) const { return reinterpret_cast<const uint32_t *>(p) + wave; }
    __device__ __forceinline__ uint32_t *col(Lbl *p) const { return reinterpret_cast<uint32_t *>(p) + wave; }
    __device__ __forceinline__ W load(uint32_t id) const { return ld_u32_global(col(words + (size_t)id * 64 + lane)); }
    __device__ __forceinline__ W load2(uint32_t lo, uint32_t hi) const {
        return ld_u32_global(col(words + (size_t)(lane < 32 ? lo : hi) * 64 + (lane & 31)));
    }
    __device__ __forceinline__ W load2h(uint32_t lo, uint32_t hi) const {
        return ld_u32_global(col(words + (size_t)(lane < 32 ? lo : hi) * 64 + 32 + (lane & 31)));
    }
    __device__ __forceinline__ W load2s(uint32_t lo, uint32_t hi, bool upper) const { return upper ? load2h(lo, hi) : load2(lo, hi); }
    __device__ __forceinline__ void store(uint32_t id, W v) { st_u32_global(col(words + (size_t)id * 64 + lane), v); }
    __device__ __forceinline__ void store2(uint32_t lo, uint32_t hi, W v) {
        st_u32_global(col(words + (size_t)(lane < 32 ? lo : hi) * 64 + (lane & 31)), v);
    }
    __device__ __forceinline__ void reveal(uint32_t slot, W v) {
        uint64_t m = __ballot(v & 1u);                                  // colour bits live in column 0
        if (wave == 0 && lane == 0) decode[slot] = m;
    }

    __device__ __forceinline__ void publish(int k, W v) { st_u32_lds(hc.sx + kSplitOp + k * kSplitWord + wave * kSplitPlane + lane, v); }
    __device__ __forceinline__ W result(int q) const { return ld_u32_lds(hc.sx + kSplitX + q * kSplitWord + wave * kSplitPlane + lane); }
    __device__ __forceinline__ void post(const SplitDesc &d, int n = 0) {
        if (wave == 0 && lane == 0) {
            typedef __attribute__((address_space(3))) gc_u32x4 *lds4;
            gc_u32x4 lo = {d.kind | ((uint32_t)n << 8), (uint32_t)d.step, (uint32_t)(d.step >> 32), (uint32_t)d.act1};
            ((lds4)(hc.sx + kSplitDesc))[0] = lo;
            if (d.kind != 3u) {
                gc_u32x4 hi = {(uint32_t)(d.act1 >> 32), (uint32_t)d.act2, (uint32_t)(d.act2 >> 32), 0u};
                ((lds4)(hc.sx + kSplitDesc))[1] = hi;
            }
        }
    }
    __device__ __forceinline__ W AND(W a, W b, uint64_t act) {
        SplitDesc d = {1u, act, 0ull, step};
        step++;
        publish(0, a);
        publish(1, b);
        post(d);
        lds_barrier();
        split_hash_phase<GARBLER>(hc, d);
        lds_barrier();
        return bit(act) ? (result(0) ^ result(1)) : 0u;
    }
    __device__ __forceinline__ void AND2(W a1, W b1, uint64_t act1, W a2, W b2, uint64_t act2, W &c1, W &c2) {
        SplitDesc d = {2u, act1, act2, step};
        step += 2;
        if (wave == 0) { SPLIT_STAMP2(0, 6) }                                       /* dual step entered (glue of the row done) */
        publish(0, a1);
        publish(1, b1);
        publish(2, a2);
        publish(3, b2);
        post(d);
        lds_barrier();
        if (wave == 0) { SPLIT_STAMP2(0, 7) }                                       /* posted, past the first barrier */
        split_hash_phase<GARBLER>(hc, d);
        if (wave == 0) { SPLIT_STAMP2(0, 8) }                                       /* own hash done, result stored */
        lds_barrier();
        if (wave == 0) { SPLIT_STAMP2(0, 9) }                                       /* past the second barrier */
        c1 = bit(act1) ? (result(0) ^ result(1)) : 0u;
        c2 = bit(act2) ? (result(2) ^ result(3)) : 0u;
        if (wave == 0) { asm volatile("" :: "v"(c1), "v"(c2)); SPLIT_STAMP2(0, 10) }  /* results read */
    }
    // x + y + carry-in over lanes [0, n) as one posted job (Circ::add picks this up): same gate steps as Circ::add_generic.
    // One hand-over barrier, then one barrier per level; after the last one the results of the last level are visible to
    // every wave, and the glue waves form the sum and the carry-out themselves, in their own layout (wave = column, lane = gate):
    //   G_fin(t) = state after the last level = G_pub(L-2)[t] ^ (bit L-1 of t ? X0 ^ X1 of level L-1 at t : 0)
    //   sum = x ^ y ^ cin ^ shl(G_fin, 1) on the active lanes;  carry out = G_fin(n - 1)
    __device__ __forceinline__ W add_native(W x, W y, int n, W cinw, W *cout) {
        const int L = sk_levels(n);
        SplitDesc d = {3u, 0ull, 0ull, step};
        const uint64_t st0 = step;
        step += 1 + (uint64_t)L;
        publish(0, x);
        publish(1, y);
        publish(2, cinw);
        post(d, n);
        lds_barrier();
        split_sk_hash<GARBLER, 0>(hc, n, st0, (const SkVals *)0);
        const lds_u32 *sx = hc.sxl;
        const int pl = wave * kSplitPlane;
        const lds_u32 *X0 = sx + kSplitKs + ((L & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;
        const lds_u32 *Gp = (L == 0) ? (sx + kSplitOp + 2 * kSplitWord + pl) : (sx + kSplitGs + (((L - 1) & 1) * 2) * kSplitWord + pl);
        // L = 0: G = X0 ^ X1 ^ cin (the first AND alone); else the published state, and the last level's gates where bit L - 1 is set
        const int tm = (lane - 1) & 63;
        const bool xm = (L == 0) || ((tm >> (L - 1)) & 1);
        const uint32_t gm = Gp[tm], x0m = X0[tm], x1m = X1[tm];
        uint32_t co = 0u;
        if (cout) {
            const bool xc = (L == 0) || (((n - 1) >> (L - 1)) & 1);
            co = Gp[n - 1] ^ (xc ? (X0[n - 1] ^ X1[n - 1]) : 0u);
        }
        const uint32_t Gsh = (lane > 0 && lane < n) ? (gm ^ (xm ? (x0m ^ x1m) : 0u)) : 0u;
        if (cout) *cout = co;
        return x ^ y ^ cinw ^ Gsh;
    }
    // the record is complete: release the hash waves
    __device__ __forceinline__ void finish() {
        SplitDesc d = {0u, 0ull, 0ull, 0ull};
        post(d);
        lds_barrier();
    }
};

// one 16-wave workgroup per record
template <bool GARBLER>
__global__ void __launch_bounds__(1024)
gc_split_kernel(const Rec *recs, uint32_t nrec, Lbl *words, Lbl *tab, uint64_t *decode, uint64_t launch_step0, Lbl R, int w, int p) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    __shared__ __attribute__((aligned(16))) uint32_t lds_sx[kSplitWords];
    lds_tab4_fill(lds_te0);
    const uint32_t wid = blockIdx.x;
    if (wid >= nrec) return;
    SplitHashCtx hc;
    hc.lt = lds_tab4_make(lds_te0);
    hc.lane = threadIdx.x & 63;
    hc.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = hc.lane & 3;
#pragma unroll
    for (int k = 0; k < 11; k++) hc.rkl[k] = c_aes.rk[4 * k + c];
    hc.Rq = GARBLER ? (c == 0 ? R.x : c == 1 ? R.y : c == 2 ? R.z : R.w) : 0u;
    hc.sx = lds_sx;
    hc.sxl = (__attribute__((address_space(3))) uint32_t *)lds_sx;
    if (threadIdx.x == 0) lds_sx[kSplitZero] = 0u;          // (visible to every wave after the first hand-over barrier)
    hc.tabw = reinterpret_cast<uint32_t *>(tab);
    hc.launch_step0 = launch_step0;
    if (hc.wave >= 4) {
        // hash waves: serve the levels the glue waves post
        const int q = hc.wave >> 2;
        // the operand words of an addition's first AND lie at fixed places: wave q = 1 reads them together with the descriptor
        const int g_ = 16 * (hc.wave & 3) + (hc.lane >> 2), pl_ = c * kSplitPlane;
        SkPtrs first1;
        sk_ptrs_own<1>(first1, hc.sxl, -1, g_, pl_);
        sk_ptrs_oth<1>(first1, hc.sxl, -1, g_, GARBLER ? 0 : pl_);
        for (;;) {
            lds_barrier();
            typedef const __attribute__((address_space(3))) gc_u32x4 *lds4;
            const gc_u32x4 lo = ((lds4)(lds_sx + kSplitDesc))[0];
            SkVals pre;
            if (q == 1) pre = sk_load<GARBLER, 1>(first1);
            SplitDesc d;
            const uint32_t kn = rfl(lo.x);
            d.kind = kn & 0xffu;
            if (d.kind == 0u) return;
            d.step = ((uint64_t)rfl(lo.z) << 32) | rfl(lo.y);
            if (d.kind == 3u) {
                // a whole addition: its last level's barrier ends the job (the glue waves pick the results up themselves)
                const int n = (int)(kn >> 8);
                if (q == 1) split_sk_hash<GARBLER, 1>(hc, n, d.step, &pre);
                else split_sk_side(hc, n);
                continue;
            }
            const gc_u32x4 hi = ((lds4)(lds_sx + kSplitDesc))[1];
            d.act1 = ((uint64_t)rfl(hi.x) << 32) | rfl(lo.w);
            d.act2 = ((uint64_t)rfl(hi.z) << 32) | rfl(hi.y);
            if (hc.wave == 4) { SPLIT_STAMP2(1, 7) }                                /* descriptor decoded */
            split_hash_phase<GARBLER>(hc, d);
            if (hc.wave == 4) { SPLIT_STAMP2(1, 8) }                                /* hashed, result stored */
            lds_barrier();
            if (hc.wave == 4) { SPLIT_STAMP2(1, 9) }
        }
    }
    SplitBackend<GARBLER> be;
    be.hc = hc;
    be.wave = hc.wave;
    be.lane = hc.lane;
    be.Rc = GARBLER ? (hc.wave == 0 ? R.x : hc.wave == 1 ? R.y : hc.wave == 2 ? R.z : R.w) : 0u;
    be.words = words;
    be.decode = decode;
    Rec r = recs[wid];
    r.op = __builtin_amdgcn_readfirstlane(r.op);
    r.cnt = __builtin_amdgcn_readfirstlane(r.cnt);
    r.dst = __builtin_amdgcn_readfirstlane(r.dst);
    r.a = __builtin_amdgcn_readfirstlane(r.a);
    r.b = __builtin_amdgcn_readfirstlane(r.b);
    r.c = __builtin_amdgcn_readfirstlane(r.c);
    r.sa = __builtin_amdgcn_readfirstlane(r.sa);
    r.sb = __builtin_amdgcn_readfirstlane(r.sb);
    uint32_t s_lo = __builtin_amdgcn_readfirstlane((uint32_t)r.step0);
    uint32_t s_hi = __builtin_amdgcn_readfirstlane((uint32_t)(r.step0 >> 32));
    be.step = ((uint64_t)s_hi << 32) | s_lo;
    exec_record(be, r, w, p);
    be.finish();
}

}  // namespace gc
