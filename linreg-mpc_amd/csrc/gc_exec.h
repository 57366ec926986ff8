// gc_exec.h -- the record format of the garbled word machine and its
// backend-generic interpreter.
//
// A *program* is a list of launches; a launch is an array of records; one
// record is executed by one wavefront (lane = bit).  Records of one launch are
// independent; launches run in order.  Garbler and evaluator execute the same
// program; the garbler writes two 16-byte ciphertexts per active lane and
// step into the launch's table buffer, the evaluator reads them back.
//
// Words live in a "word file": word id -> 64 labels (1 KiB, lane-major), so a
// wave loads/stores a word as one coalesced 1 KiB access (global_load_dwordx4).
#pragma once
#include <assert.h>
#include <stdint.h>

#include "gc_circuits.h"

namespace gc {

enum Op : uint32_t {
    OP_NOP = 0,
    OP_MAC,     // (S,C) = sum_{k<cnt} mul(a+k*sa, b+k*sb), carry-save; S->dst, C->dst+1
    OP_SUM,     // dst = sum_{k<cnt} words[a+k*sa]                      (mod 2^w)
    OP_SUBSUM,  // dst = words[c] - sum_{k<cnt} words[a+k*sa]
    OP_IPMAC,   // IpAcc = sum_{k<cnt} (a+k*sa)*(b+k*sb) exact; 4 words -> dst..dst+3
    OP_IPFIN,   // dst = wrap((sum of cnt IpAcc at a+4k) >> p)
    OP_IPMERGE, // IpAcc at dst..dst+3 = sum of cnt IpAcc at a+4k
    OP_MUL,     // dst = mul(a, b); cnt = 2: also words[dst + sa] = hdiff(dst) (the record that makes a Karatsuba operand final forms its half differences)
    OP_MULSUB,  // dst = words[c] - mul(a, b); cnt = 2: also words[dst + sa] = |dst|; cnt = 3: words[dst + sa] = hdiff(dst)
    OP_ADD,     // dst = a + b
    OP_SUB,     // dst = a - b
    OP_ABS,     // dst = |a|
    OP_MAX,     // dst = max_{k<cnt} words[a+k*sa]   (ordering of Circ::gt)
    OP_DIV,     // dst = div(a, b); c != 0: words[c] = dst too (the mirrored entry of a symmetric matrix); cnt = 2: words[dst + sa] = hdiff(dst)
    OP_SQRT,    // dst = sqrt(a)
    OP_IDIVC,   // dst = tdiv(a, public constant c)   (linear.oc:52-65, normalizer)
    OP_CONST,   // dst = public constant (a = low 32 bits, b = high 32 bits)
    OP_COPY,    // dst = a
    OP_REVEAL,  // decode[dst] = colour bits of word a (cnt unused)
    OP_MAC2,    // 32-bit only: two OP_MAC chunks per wave; products k (lanes 0..31) and cnt+k (lanes 32..63);
                // (S,C) of the first chunk -> dst, dst+1, of the second -> dst+2, dst+3
    OP_MACK,    // 64-bit only: OP_MAC through the Karatsuba circuit (Circ::mack2), two products at a time; the words
                // hdiff(x) of both operand vectors lie c words above the operands: (a + k*sa) + c, (b + k*sb) + c
    OP_HDIFF,   // dst = hdiff(a): |hi32(a) - lo32(a)| and its sign, for OP_MACK
    OP_EQ,      // dst = [a == b] in lane 0 (other lanes 0): the comparison of the two parties' dimensions, src/linear.oc:109-114
    OP_DIVB,    // dst = div(a, b) where the program guarantees |a| <= |b| (w = 64 only): p + 1 quotient bits, Circ::div_mag
    OP_COUNT_
};

struct Rec {
    uint32_t op, cnt;
    uint32_t dst, a, b, c;
    int32_t sa, sb;
    uint64_t step0;   // global index of this record's first gate step
};

// the record dst = tdiv(a, c) for a public c > 0 with the constants of Circ::divc: m = floor(2^(w-1+l) / c) + 1,
// l = ceil(log2 c) (host side: a 128-bit division)
// c >= 1: the only caller divides by the system's dimension (check_system rejects d < 1); c = 0 has no meaning here
inline Rec idivc_rec(uint32_t dst, uint32_t a, uint32_t c, int w) {
    assert(c >= 1);
    int l = 0;
    while (l < 32 && (1ull << l) < (uint64_t)c) l++;
    const uint64_t m = c > 1 ? (uint64_t)((((unsigned __int128)1) << (w - 1 + l)) / c) + 1 : 0;
    Rec r;
    r.op = OP_IDIVC; r.cnt = 1; r.dst = dst; r.a = a; r.b = (uint32_t)m; r.c = c; r.sa = l; r.sb = (int32_t)(uint32_t)(m >> 32);
    r.step0 = 0;
    return r;
}

// number of gate steps / active AND gates of one record (host side)
inline void rec_cost(const Rec &r, int w, int p, uint64_t &steps, uint64_t &gates, uint64_t *xors);

// Backend extras required here: load(id) / store(id, W) / reveal(slot, W).
template <class B>
GC_HD void exec_record(B &be, const Rec &r, int w, int p) {
    typedef Circ<B> C;
    typedef typename B::W W;
    switch (r.op) {
    case OP_MAC: {
        W S = be.zero(), Cc = be.zero();
        for (uint32_t k = 0; k < r.cnt; k++)
            C::mac(be, S, Cc, be.load(r.a + (int32_t)k * r.sa), be.load(r.b + (int32_t)k * r.sb), w, p);
        be.store(r.dst, S);
        be.store(r.dst + 1, Cc);
    } break;
    case OP_MAC2: {
        W S = be.zero(), Cc = be.zero();
        for (uint32_t k = 0; k < r.cnt; k++)
            C::mac2(be, S, Cc, be.load2(r.a + (int32_t)k * r.sa, r.a + (int32_t)(r.cnt + k) * r.sa),
                    be.load2(r.b + (int32_t)k * r.sb, r.b + (int32_t)(r.cnt + k) * r.sb), p);
        be.store2(r.dst, r.dst + 2, S);
        be.store2(r.dst + 1, r.dst + 3, Cc);
    } break;
    case OP_MACK: {
        W S = be.zero(), Cc = be.zero();
        C::mack_rec(be, S, Cc, r.a, r.b, r.sa, r.sb, r.cnt, r.c, p);
        be.store(r.dst, S);
        be.store(r.dst + 1, Cc);
    } break;
    case OP_HDIFF:
        be.store(r.dst, C::hdiff(be, be.load(r.a)));
        break;
    case OP_EQ:
        be.store(r.dst, C::equal(be, be.load(r.a), be.load(r.b), w));
        break;
    case OP_SUM:
    case OP_SUBSUM: {
        W S = be.load(r.a), Cc = be.zero();
        for (uint32_t k = 1; k < r.cnt; k++) C::csa(be, S, Cc, be.load(r.a + (int32_t)k * r.sa), w);
        W v = (r.cnt > 1) ? C::add(be, S, Cc, w) : S;
        if (r.op == OP_SUBSUM) v = C::sub(be, be.load(r.c), v, w);
        be.store(r.dst, v);
    } break;
    case OP_IPMAC: {
        typename C::IpAcc A;
        C::ip_zero(be, A);
        for (uint32_t k = 0; k < r.cnt; k++)
            C::ip_mac(be, A, be.load(r.a + (int32_t)k * r.sa), be.load(r.b + (int32_t)k * r.sb), w, p);
        be.store(r.dst, A.LS); be.store(r.dst + 1, A.LC);
        be.store(r.dst + 2, A.HS); be.store(r.dst + 3, A.HC);
    } break;
    case OP_IPFIN:
    case OP_IPMERGE: {
        typename C::IpAcc A, O;
        A.LS = be.load(r.a); A.LC = be.load(r.a + 1); A.HS = be.load(r.a + 2); A.HC = be.load(r.a + 3);
        for (uint32_t k = 1; k < r.cnt; k++) {
            uint32_t o = r.a + 4 * k;
            O.LS = be.load(o); O.LC = be.load(o + 1); O.HS = be.load(o + 2); O.HC = be.load(o + 3);
            C::ip_merge(be, A, O, w, p);
        }
        if (r.op == OP_IPFIN) {
            be.store(r.dst, C::ip_final(be, A, w, p));
        } else {
            be.store(r.dst, A.LS); be.store(r.dst + 1, A.LC);
            be.store(r.dst + 2, A.HS); be.store(r.dst + 3, A.HC);
        }
    } break;
    case OP_MUL: {
        W v = C::mul(be, be.load(r.a), be.load(r.b), w, p);
        be.store(r.dst, v);
        if (r.cnt == 2) be.store(r.dst + (uint32_t)r.sa, C::hdiff(be, v));
    } break;
    case OP_MULSUB: {
        W X, Y;
        C::mul_xy(be, be.load(r.a), be.load(r.b), w, p, X, Y);
        W v = C::add(be, X, Y, w);
        v = C::sub(be, be.load(r.c), v, w);
        be.store(r.dst, v);
        // (end of round 5: what the next launch would have formed from this word alone, in the record that makes it -- one
        // dependent launch less per CGD iteration each: |g_i| for the maximum, hdiff(p_i) for the Karatsuba products)
        if (r.cnt == 2) be.store(r.dst + (uint32_t)r.sa, C::vabs(be, v, w));
        else if (r.cnt == 3) be.store(r.dst + (uint32_t)r.sa, C::hdiff(be, v));
    } break;
    case OP_ADD:
        be.store(r.dst, C::add(be, be.load(r.a), be.load(r.b), w));
        break;
    case OP_SUB:
        be.store(r.dst, C::sub(be, be.load(r.a), be.load(r.b), w));
        break;
    case OP_ABS:
        be.store(r.dst, C::vabs(be, be.load(r.a), w));
        break;
    case OP_MAX: {
        W m = be.load(r.a);
        for (uint32_t k = 1; k < r.cnt; k++) m = C::vmax(be, be.load(r.a + (int32_t)k * r.sa), m, w);
        be.store(r.dst, m);
    } break;
    case OP_DIV: {
        // (round 5: the quotient's mirror and its half-difference word in the record that makes it final -- as launches of their
        // own, OP_COPY / OP_HDIFF were one more dependent launch per column of a factorisation, each waiting for CUs beside
        // the other role's MAC kernel)
        W v = C::div(be, be.load(r.a), be.load(r.b), w, p);
        be.store(r.dst, v);
        if (r.c) be.store(r.c, v);
        if (r.cnt == 2) be.store(r.dst + (uint32_t)r.sa, C::hdiff(be, v));
    } break;
    case OP_DIVB:
        be.store(r.dst, C::div(be, be.load(r.a), be.load(r.b), w, p, true));
        break;
    case OP_SQRT:
        be.store(r.dst, C::vsqrt(be, be.load(r.a), w, p));
        break;
    case OP_IDIVC:
        // c: the divisor; b | sb << 32: its multiplier m and sa: l = ceil(log2 c) (idivc_rec below)
        be.store(r.dst, C::divc(be, be.load(r.a), (uint64_t)r.b | ((uint64_t)(uint32_t)r.sb << 32), (int)r.sa, w));
        break;
    case OP_CONST:
        be.store(r.dst, be.sel(lanes(w), be.konst((uint64_t)r.a | ((uint64_t)r.b << 32)), be.zero()));
        break;
    case OP_COPY:
        be.store(r.dst, be.load(r.a));
        break;
    case OP_REVEAL:
        be.reveal(r.dst, be.load(r.a));
        break;
    default:
        break;
    }
}

// PlainBackend with a word file: host-side cost model and plaintext checks
struct PlainMachine : PlainBackend {
    uint64_t *words;
    uint64_t *decode;
    PlainMachine(uint64_t *w_, uint64_t *d_) : words(w_), decode(d_) {}
    W load(uint32_t id) const { return words[id]; }
    void store(uint32_t id, W v) { words[id] = v; }
    // lanes 0..31 of word lo | lanes 0..31 of word hi moved to lanes 32..63 (and back)
    W load2(uint32_t lo, uint32_t hi) const { return (words[lo] & 0xffffffffull) | (words[hi] << 32); }
    // lanes 32..63 of word lo in lanes 0..31 | lanes 32..63 of word hi
    W load2h(uint32_t lo, uint32_t hi) const { return (words[lo] >> 32) | (words[hi] & 0xffffffff00000000ull); }
    W load2s(uint32_t lo, uint32_t hi, bool upper) const { return upper ? load2h(lo, hi) : load2(lo, hi); }
    void store2(uint32_t lo, uint32_t hi, W v) { words[lo] = v & 0xffffffffull; words[hi] = v >> 32; }
    void reveal(uint32_t slot, W v) { if (decode) decode[slot] = v; }
};

// cost of a record: run it on a scratch plaintext machine (the circuits'
// control flow is data-independent, so any operand values give the counts)
// xors (optional): XOR gates a flat gate list of the same circuit would hold -- every word-level XOR of two wire words
// counted as w gates (lane moves, selections by public masks, constants and inverters are wiring): the figure behind
// SURVEY.md 8(d)'s flat-list traffic formula 192 N_AND + 128 N_XOR, which this engine's word machine never pays
inline void rec_cost(const Rec &r, int w, int p, uint64_t &steps, uint64_t &gates, uint64_t *xors = 0) {
    struct CostMachine : PlainBackend {
        uint64_t nx = 0;
        W XOR(W a, W b) { nx++; return a ^ b; }
        W load(uint32_t) const { return 0x5a5a5a5a5a5a5a5aull; }
        W load2(uint32_t, uint32_t) const { return 0x5a5a5a5a5a5a5a5aull; }
        W load2h(uint32_t, uint32_t) const { return 0x5a5a5a5a5a5a5a5aull; }
        W load2s(uint32_t, uint32_t, bool) const { return 0x5a5a5a5a5a5a5a5aull; }
        void store(uint32_t, W) {}
        void store2(uint32_t, uint32_t, W) {}
        void reveal(uint32_t, W) {}
    } m;
    exec_record(m, r, w, p);
    steps = m.steps;
    gates = m.gates;
    if (xors) *xors = m.nx * (uint64_t)w;
}

}  // namespace gc
