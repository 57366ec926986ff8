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
__device__ __forceinline__ uint32_t hash_split(const LdsTab4 &lt, const uint32_t rkl[11], uint32_t x, uint32_t twc, int c) {
    const uint32_t t = quad_perm<0x4E>(x);                 // lanes 0,1 <- columns 2,3 ; lanes 2,3 <- columns 0,1
    const uint32_t k = (c < 2) ? (t ^ twc) : (x ^ t);
    uint32_t s = k ^ rkl[0];
#ifdef GC_X_NOHASH          /* timing experiments only (scripts/exp): results are wrong with it */
    return s;
#endif
#pragma unroll
    for (int rnd = 1; rnd < 10; rnd++) {
        // (looking up with the own column and routing the results as DPP operands of the XORs saves the three moves but
        // measured no faster: scripts/exp/lat4.hip 2 118 against 2 142 cycles, the divider 2 % slower)
        const uint32_t s1 = quad_perm<0x39>(s), s2 = quad_perm<0x4E>(s), s3 = quad_perm<0x93>(s);   // columns c+1, c+2, c+3
        const uint32_t v0 = lt.lkt(0, s, 0), v1 = lt.lkt(1, s1, 1), v2 = lt.lkt(2, s2, 2), v3 = lt.lkt(3, s3, 3);
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
    kSplitDesc = 20 * kSplitWord,          // kind, act1 (2), act2 (2), step (2): two 16-byte stores
    kSplitWords = 20 * kSplitWord + 8
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
// ONE barrier per level.  Posted by the glue waves with x, y and the carry-in word in operand words 0..2; sum and final
// generate word come back in result words 0 and 1.  Gate steps, their order, lanes and activity masks are exactly those of
// Circ::add_generic: the first AND, then one step per level k, in which lane i with bit k set holds the G-gate of node i
// (P_i & G_m, m = the top lane of the lower half of i's block of 2^(k+1)) and lane i - 2^k the P-gate of node i (P_i & P_m).
//
// A level is a single gate step: two hashes per gate, waves q = 0 (operand a) and q = 1 (operand b), eight of the sixteen
// waves -- 1484 cycles in scripts/exp/lat4.hip against 2142 for the four-hash levels of the Kogge-Stone form this replaces
// (round 2), and six levels after the first AND in both.  Wave q = 2 keeps the published state one level behind:
//   state after level k - 1 at lane t:  G = G_pub(k-2)[t] ^ (bit k-1 of t ? X0 ^ X1 of level k-1 at t : 0)
//                                       P = (t was a P-node of level k-1 ? X0 ^ X1 of level k-1 at t - 2^(k-1) : P_pub(k-2)[t])
// (X0, X1: the two hash results of a gate; their XOR is the gate's output label).  Every hash lane rebuilds the two
// states it needs -- its own node's and m's -- from what the last barrier made visible, the publisher writes state k - 1
// for level k + 1; results and published states alternate between two buffers.
#ifndef GC_SPLIT_TRACE
#define GC_SPLIT_TRACE 0      /* timing experiments only (scripts/exp/split_trace.py): s_memtime stamps of the adder levels of record 0 */
#endif
#if GC_SPLIT_TRACE
static __device__ uint64_t g_split_trace[2 * 8192];
static __device__ uint32_t g_split_trace_n[2];
#define SPLIT_STAMP(tag)                                                                                         \
    if (tr_on) {                                                                                                 \
        uint64_t t_;                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                                         \
        if (hc.lane == 0 && tr_i < 8192) g_split_trace[tr_w * 8192 + tr_i] = (t_ << 4) | (uint64_t)(tag);        \
        tr_i++;                                                                                                  \
    }
#else
#define SPLIT_STAMP(tag)
#endif
template <bool GARBLER>
__device__ __forceinline__ void split_sk_add(const SplitHashCtx &hc, const SplitDesc &d) {
    const int q = hc.wave >> 2, r = hc.wave & 3, c = hc.lane & 3, g = 16 * r + (hc.lane >> 2);
#if GC_SPLIT_TRACE
    const bool tr_on = blockIdx.x == 0 && (hc.wave == 0 || hc.wave == 4);
    const int tr_w = hc.wave == 4 ? 1 : 0;
    uint32_t tr_i = tr_on ? g_split_trace_n[tr_w] : 0u;
    SPLIT_STAMP(1)                                                                  /* addition entered (after the hand-over barrier) */
#endif
    const int pl = c * kSplitPlane;
    const uint64_t act = d.act1;
    const int n = 64 - __builtin_clzll(act);                                    // act = lanes(n)
    const uint32_t *opx = hc.sx + kSplitOp + pl, *opy = opx + kSplitWord, *opc = opx + 2 * kSplitWord;
    uint64_t st = d.step;
    // evaluator: the ciphertext row of a level is fetched one level ahead (every step index of the addition is known)
    uint32_t tg = 0, tgn = 0;
    if (!GARBLER && q < 2) {
        tg = ld_u32_global(split_row(hc, q, st));
        if (n > 1) tgn = ld_u32_global(split_row(hc, q, st + 1));
    }
    // first AND: G = ((x ^ cin) & (y ^ cin)) ^ cin.  Results of level k go to buffer (k + 1) & 1; this step counts as level -1
    if (q < 2) {
        const uint32_t xg = ld_u32_lds(opx + g), yg = ld_u32_lds(opy + g), cg = ld_u32_lds(opc + g);
        split_hash_core<GARBLER>(hc, q, act, st, xg ^ cg, yg ^ cg, hc.sx + kSplitKs + q * kSplitWord, tg);
    }
    st += 1;
    SPLIT_STAMP(2)                                                                  /* first AND hashed, result stored */
    lds_barrier();
    SPLIT_STAMP(3)                                                                  /* past the barrier */
    int k = 0;
    for (int h = 1; h < n; h <<= 1, k++) {
        const uint32_t *X0 = hc.sx + kSplitKs + ((k & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;   // results of level k - 1
        const uint32_t *Gp = hc.sx + kSplitGs + (((k - 1) & 1) * 2) * kSplitWord + pl, *Pp = Gp + kSplitWord;   // published state k - 2
        const int hp = h >> 1;
        const uint64_t bitk = bit_lanes(k) & act;
        const uint64_t pn = ((2 * h) < n) ? (bitk & ~(((2 * h) >= 64) ? ~0ull : ((1ull << (2 * h)) - 1ull))) : 0ull;
        const uint64_t stepact = bitk | (pn >> h);
        tg = tgn;
        if (!GARBLER && q < 2 && (2 * h) < n) tgn = ld_u32_global(split_row(hc, q, st + 1));     // row of the next level
        // The state after level k - 1 at the two lanes this lane needs, WITHOUT branches (every load is issued before the
        // first wait; a divergent if / else per role and per case cost two to four LDS round trips per level):
        //   own node t = g (G-gate lane) or g + h (P-gate host):   P
        //   m = top lane of the lower half of the block:           G (G-gate lane) or P (host)
        // level 0 reads the first AND's results and the operand words; later levels the published state and the last results
        if (q < 2) {
            const bool up = (bitk >> g) & 1ull, host = ((pn >> h) >> g) & 1ull;
            const int mb = (g & ~(2 * h - 1)) | (h - 1);
            const int t = up ? g : ((g + h) & 63);
            uint32_t a_op, b_op;
            if (k == 0) {
                const uint32_t px = ld_u32_lds(opx + t), py = ld_u32_lds(opy + t);
                const uint32_t m0 = ld_u32_lds(X0 + mb), m1 = ld_u32_lds(X1 + mb), mc = ld_u32_lds(opc + mb);
                const uint32_t mx = ld_u32_lds(opx + mb), my = ld_u32_lds(opy + mb);
                a_op = px ^ py;
                b_op = up ? (m0 ^ m1 ^ mc) : (mx ^ my);
            } else {
                const bool tp = ((t >> (k - 1)) & 1) && t >= h;                  // t was a P-node of level k - 1
                const int ti = tp ? t - hp : t;
                const uint32_t p1 = ld_u32_lds((tp ? X0 : Pp) + ti), p2 = ld_u32_lds(X1 + ti);
                const int mi = up ? mb : ((mb - hp) & 63);                     // (m is a G-node and, for hosts, a P-node of level k - 1)
                const uint32_t m0 = ld_u32_lds(X0 + mi), m1 = ld_u32_lds(X1 + mi), mg = ld_u32_lds(Gp + mb);
                a_op = tp ? (p1 ^ p2) : p1;
                b_op = m0 ^ m1 ^ (up ? mg : 0u);
            }
            if (!(up || host)) { a_op = 0u; b_op = 0u; }
#if GC_SPLIT_TRACE
            if (tr_on) { asm volatile("" :: "v"(a_op), "v"(b_op)); }
            SPLIT_STAMP(4)                                                          /* operands rebuilt (LDS reads back) */
#endif
            split_hash_core<GARBLER>(hc, q, stepact, st, a_op, b_op, hc.sx + kSplitKs + (((k + 1) & 1) * 4 + q) * kSplitWord, tg);
            SPLIT_STAMP(5)                                                          /* hashed, result stored */
        } else if (q == 2) {
            // publish state k - 1 of the own lane (read at level k + 1)
            uint32_t *Go = hc.sx + kSplitGs + ((k & 1) * 2) * kSplitWord + pl;
            const uint32_t x0 = ld_u32_lds(X0 + g), x1 = ld_u32_lds(X1 + g);
            uint32_t Gs, Ps;
            if (k == 0) {
                Gs = x0 ^ x1 ^ ld_u32_lds(opc + g);
                Ps = ld_u32_lds(opx + g) ^ ld_u32_lds(opy + g);
            } else {
                const bool gp = ((g >> (k - 1)) & 1) && g >= h;
                const int gi = gp ? g - hp : g;
                const uint32_t p1 = ld_u32_lds((gp ? X0 : Pp) + gi), p2 = ld_u32_lds(X1 + gi), go = ld_u32_lds(Gp + g);
                Gs = go ^ (((g >> (k - 1)) & 1) ? (x0 ^ x1) : 0u);
                Ps = gp ? (p1 ^ p2) : p1;
            }
            st_u32_lds(Go + g, Gs);
            st_u32_lds(Go + kSplitWord + g, Ps);
        }
        st += 1;
        lds_barrier();
        SPLIT_STAMP(3)
    }
#if GC_SPLIT_TRACE
    if (tr_on && hc.lane == 0) g_split_trace_n[tr_w] = tr_i < 8192 ? tr_i : 8192;
#endif
    // carries = shl(G, 1) ^ cin on the active lanes; sum = P ^ carries; the final generate word for the carry out.
    // Final G at lane t: the state after the last level (k levels were run)
    if (q == 0) {
        const uint32_t *X0 = hc.sx + kSplitKs + ((k & 1) * 4) * kSplitWord + pl, *X1 = X0 + kSplitWord;
        const uint32_t *Gp = hc.sx + kSplitGs + (((k - 1) & 1) * 2) * kSplitWord + pl;
        auto G_fin = [&](int t) -> uint32_t {
            const uint32_t xr = ld_u32_lds(X0 + t) ^ ld_u32_lds(X1 + t);
            if (k == 0) return xr ^ ld_u32_lds(opc + t);
            return ld_u32_lds(Gp + t) ^ (((t >> (k - 1)) & 1) ? xr : 0u);
        };
        const uint32_t P0 = ld_u32_lds(opx + g) ^ ld_u32_lds(opy + g), cg = ld_u32_lds(opc + g);
        const bool in = g < n;
        const uint32_t Gsh = (in && g > 0) ? G_fin(g - 1) : 0u;
        st_u32_lds(hc.sx + kSplitX + pl + g, P0 ^ Gsh ^ cg);
        st_u32_lds(hc.sx + kSplitX + kSplitWord + pl + g, in ? G_fin(g) : 0u);
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
    __device__ __forceinline__ W shl(W a, int k) const { return k >= 64 ? 0u : pull(a, lane - k, lane >= k); }
    __device__ __forceinline__ W shr(W a, int k) const { return k >= 64 ? 0u : pull(a, lane + k, lane + k < 64); }

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
    __device__ __forceinline__ void post(const SplitDesc &d) {
        if (wave == 0 && lane == 0) {
            typedef __attribute__((address_space(3))) gc_u32x4 *lds4;
            gc_u32x4 lo = {d.kind, (uint32_t)d.act1, (uint32_t)(d.act1 >> 32), (uint32_t)d.act2};
            gc_u32x4 hi = {(uint32_t)(d.act2 >> 32), (uint32_t)d.step, (uint32_t)(d.step >> 32), 0u};
            ((lds4)(hc.sx + kSplitDesc))[0] = lo;
            ((lds4)(hc.sx + kSplitDesc))[1] = hi;
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
        publish(0, a1);
        publish(1, b1);
        publish(2, a2);
        publish(3, b2);
        post(d);
        lds_barrier();
        split_hash_phase<GARBLER>(hc, d);
        lds_barrier();
        c1 = bit(act1) ? (result(0) ^ result(1)) : 0u;
        c2 = bit(act2) ? (result(2) ^ result(3)) : 0u;
    }
    // x + y + carry-in over lanes [0, n) as one posted job (Circ::add picks this up): same gate steps as Circ::add_generic,
    // two barriers around the whole addition plus one per level instead of two per level
    __device__ __forceinline__ W add_native(W x, W y, int n, W cinw, W *cout) {
        const uint64_t act = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
        SplitDesc d = {3u, act, 0ull, step};
        step += 1;
        for (int h = 1; h < n; h <<= 1) step += 1;
        publish(0, x);
        publish(1, y);
        publish(2, cinw);
        post(d);
        lds_barrier();
        split_sk_add<GARBLER>(hc, d);
        lds_barrier();
        if (cout) *cout = bcast(result(1), n - 1);
        return result(0);
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
    hc.tabw = reinterpret_cast<uint32_t *>(tab);
    hc.launch_step0 = launch_step0;
    if (hc.wave >= 4) {
        // hash waves: serve the levels the glue waves post
        for (;;) {
            lds_barrier();
            typedef const __attribute__((address_space(3))) gc_u32x4 *lds4;
            const gc_u32x4 lo = ((lds4)(lds_sx + kSplitDesc))[0];
            SplitDesc d;
            d.kind = rfl(lo.x);
            if (d.kind == 0u) return;
            const gc_u32x4 hi = ((lds4)(lds_sx + kSplitDesc))[1];
            d.act1 = ((uint64_t)rfl(lo.z) << 32) | rfl(lo.y);
            d.act2 = ((uint64_t)rfl(hi.x) << 32) | rfl(lo.w);
            d.step = ((uint64_t)rfl(hi.z) << 32) | rfl(hi.y);
            if (d.kind == 3u) split_sk_add<GARBLER>(hc, d);
            else split_hash_phase<GARBLER>(hc, d);
            lds_barrier();
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
