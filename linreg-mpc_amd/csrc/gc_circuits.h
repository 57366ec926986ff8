// gc_circuits.h -- wave-shaped boolean circuits for the fixed-point ops.
//
// One "word" W is one wire per lane of a 64-wide wavefront: lane l carries bit
// l of a two's-complement value (w = 32 or 64 lanes used).  Circuits are
// written once, against an abstract backend B, as sequences of whole-wave
// steps; every B::AND call is one *gate step* (up to 64 AND gates, one per
// active lane); B::AND2 is two independent gate steps issued together.  XOR / NOT / lane moves / public constants are free (free-XOR).
//
// Backends (all expose the same members):
//   PlainBackend            W = uint64_t bit mask; used on the host to count
//                           steps / gates and (tests) to check circuit logic
//   GpuGarbler/GpuEvaluator W = one 128-bit label per lane  (gc_device.h)
//   CpuGarbler/CpuEvaluator oracle/gc_cpu.cpp (CPU baseline, AES-NI)
//
// Semantics implemented (reference file:line):
//   add/sub   src/fixed.oc:99-121      wrap_w(a +- b)
//   abs       src/fixed.oc:90-97
//   gt        src/fixed.oc:78-88       (unsigned for w=64, signed for w=32)
//   mul       src/fixed.oc:149-162     wrap_w((a*b) >> p), exact product
//   ip        src/fixed.oc:124-147     wrap_w((sum a_i*b_i) >> p)
//   div       src/fixed.oc:164-188     wrap_w(tdiv(a << p, b))
//   sqrt      src/fixed.oc:217-248
// The bit-level structure (carry-save array multiplier, Sklansky adders,
// restoring divider) is this build's own; only the integer results are the
// reference's.
#pragma once
#include <stdint.h>

#ifndef GC_HD
#if defined(__HIPCC__)
#define GC_HD __host__ __device__ __forceinline__
#else
#define GC_HD inline
#endif
#endif

namespace gc {

GC_HD uint64_t lanes(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1)); }
// lanes whose index has bit k set (a select chain: an indexed constant table would be a memory load per adder level)
GC_HD uint64_t bit_lanes(int k) {
    return k == 0 ? 0xaaaaaaaaaaaaaaaaull : k == 1 ? 0xccccccccccccccccull : k == 2 ? 0xf0f0f0f0f0f0f0f0ull
         : k == 3 ? 0xff00ff00ff00ff00ull : k == 4 ? 0xffff0000ffff0000ull : 0xffffffff00000000ull;
}

template <class B>
struct Circ {
    typedef typename B::W W;

    // ---- Kogge-Stone adder over lanes [0, n): returns x + y + cin (mod 2^n).
    // cinw: word whose lane 0 holds the carry-in (other lanes zero) or zero().
    // cout (optional): carry out of lane n-1, broadcast to every lane.
    static GC_HD W add(B &be, W x, W y, int n, W cinw, W *cout) { return add_pick(be, x, y, n, cinw, cout, 0); }
    // a backend may run the whole addition itself (same gate steps in the same order): gc_split.h
    template <class BB>
    static GC_HD auto add_pick(BB &be, W x, W y, int n, W cinw, W *cout, int) -> decltype(be.add_native(x, y, n, cinw, cout)) {
        return be.add_native(x, y, n, cinw, cout);
    }
    template <class BB>
    static GC_HD W add_pick(BB &be, W x, W y, int n, W cinw, W *cout, long) { return add_generic(be, x, y, n, cinw, cout); }
    // Sklansky prefix adder.  Level k merges, in every block of 2^(k+1) lanes, the upper half with the top node m of the
    // lower half:  G_i ^= P_i & G_m,  P_i &= P_m  for the lanes i with bit k set.  That is n/2 nodes of two ANDs each --
    // n gates, ONE 64-lane gate step per level: the G-gate of node i sits in lane i, its P-gate in lane i - 2^k (a lane
    // of the lower half, idle at this level).  1 + ceil(log2 n) gate steps for an n-bit addition (64 bits: 7; the
    // Kogge-Stone form of rounds 1-2 -- n nodes per level, two steps -- took 12), the same 1 + log2 n dependent levels,
    // and every level is a SINGLE step (two hashes per gate on the critical path instead of four).  The fan-out of node m
    // to its block is a lane move (B::bblk), free like every other wire permutation.  P_i is dropped once i's prefix
    // reaches lane 0 (i < 2^(k+1)) and on the last level.
    static GC_HD W add_generic(B &be, W x, W y, int n, W cinw, W *cout) {
        const uint64_t act = lanes(n);
        W P = be.XOR(x, y);
        // lane 0: maj(x0, y0, cin) = ((x0^c)&(y0^c))^c ; other lanes: x&y
        W G = be.XOR(be.AND(be.XOR(x, cinw), be.XOR(y, cinw), act), cinw);
        W Pg = P;
        for (int k = 0; (1 << k) < n; k++) {
            const int h = 1 << k;
            const uint64_t bitk = bit_lanes(k) & act;                               // nodes of this level
            const uint64_t pn = ((2 * h) < n) ? (bitk & ~lanes(2 * h)) : 0ull;       // ... whose P is still needed
            const uint64_t host = pn >> h;
            W Gm = be.bblk(G, k), Pm = be.bblk(Pg, k);
            W t = be.AND(be.sel(bitk, Pg, be.shr(Pg, h)), be.sel(bitk, Gm, Pm), bitk | host);
            G = be.XOR(G, be.sel(bitk, t, be.zero()));
            Pg = be.sel(pn, be.shl(t, h), Pg);
        }
        if (cout) *cout = be.bcast(G, n - 1);
        W carries = be.XOR(be.sel(act, be.shl(G, 1), be.zero()), cinw);
        return be.XOR(P, carries);
    }
    static GC_HD W add(B &be, W x, W y, int n) { return add(be, x, y, n, be.zero(), (W *)0); }
    // x - y (mod 2^n); cout = 1 iff x >= y (unsigned)
    static GC_HD W sub(B &be, W x, W y, int n, W *cout) {
        return add(be, x, be.NOTm(y, lanes(n)), n, be.konst(1), cout);
    }
    static GC_HD W sub(B &be, W x, W y, int n) { return sub(be, x, y, n, (W *)0); }

    // (x ^ s) + s with s a broadcast bit: conditional negate
    static GC_HD W condneg(B &be, W x, W sb, int n) {
        W xs = be.XOR(x, be.sel(lanes(n), sb, be.zero()));
        return add(be, xs, be.zero(), n, be.sel(1ull, sb, be.zero()), (W *)0);
    }
    static GC_HD W vabs(B &be, W a, int w) { return condneg(be, a, be.bcast(a, w - 1), w); }

    // sel ? a : b  (selb broadcast)
    static GC_HD W mux(B &be, W selb, W a, W b, int n) {
        return be.XOR(b, be.AND(selb, be.XOR(a, b), lanes(n)));
    }
    // a > b as a broadcast bit.  w = 64: unsigned (obig_cmp); w = 32: signed.
    static GC_HD W gt(B &be, W a, W b, int w) {
        if (w == 32) { a = be.NOTm(a, 1ull << 31); b = be.NOTm(b, 1ull << 31); }
        W ge_ba;
        (void)sub(be, b, a, w, &ge_ba);           // b >= a
        return be.NOTm(ge_ba, ~0ull);
    }
    static GC_HD W vmax(B &be, W a, W b, int w) { return mux(be, gt(be, a, b, w), a, b, w); }

    // carry-save step: (S, C) += X over n lanes (mod 2^n).  cin_lane0 is XORed
    // into lane 0 of the new carry word (always free); cout gets the carry out
    // of lane n-1 placed in lane 0 (others zero) when requested.
    static GC_HD void csa(B &be, W &S, W &C, W X, int n, W cin_lane0, W *cout) {
        const uint64_t act = lanes(n);
        W t = be.AND(be.XOR(S, X), be.XOR(C, X), act);
        W carry = be.XOR(t, be.sel(act, X, be.zero()));
        S = be.XOR(be.XOR(S, C), X);
        if (cout) *cout = be.sel(1ull, be.shr(carry, n - 1), be.zero());
        C = be.XOR(be.sel(act, be.shl(carry, 1), be.zero()), cin_lane0);
    }
    static GC_HD void csa(B &be, W &S, W &C, W X, int n) { csa(be, S, C, X, n, be.zero(), (W *)0); }

    // ---- signed w x w carry-save array multiplier, product kept mod 2^(w+p).
    //   L      : product bits 0..w-1 (resolved)
    //   S, C   : S + C = product bits w..w+p-1 (lanes 0..p-1, others zero)
    // Baugh-Wooley sign handling: partial products of the last row / last lane
    // are inverted and the constant 2^w is injected as an initial carry.
    static GC_HD void mul_core(B &be, W a, W b, int w, int p, W &L, W &S, W &C) {
        const int M = w + p;
        S = be.zero();
        L = be.zero();
        C = (p >= 1) ? be.konst(1ull << (w - 1)) : be.zero();
        if (B::kPairSteps) {
            // latency-bound backends: the partial product of row r + 1 does not depend on row r, so it
            // issues together with the carry-save AND of row r (one dual step): w levels instead of 2w - 1
            W pp = be.AND(a, be.bcast(b, 0), lanes(w < M ? w : M));
            for (int r = 0; r < w; r++) {
                const int na = (M - r) < w ? (M - r) : w;
                const uint64_t act = lanes(na);
                uint64_t inv = (r == w - 1) ? lanes(w - 1) : (1ull << (w - 1));
                pp = be.NOTm(pp, inv & act);
                W ppn = be.zero();
                const int nan = (M - r - 1) < w ? (M - r - 1) : w;
                if (r == 0) {
                    S = pp;
                    if (w > 1) ppn = be.AND(a, be.bcast(b, 1), lanes(nan));
                } else {
                    W t;
                    if (r + 1 < w) be.AND2(be.XOR(S, pp), be.XOR(C, pp), act, a, be.bcast(b, r + 1), lanes(nan), t, ppn);
                    else t = be.AND(be.XOR(S, pp), be.XOR(C, pp), act);
                    W Sn = be.XOR(be.XOR(S, C), pp);
                    C = be.XOR(t, pp);
                    S = Sn;
                }
                L = be.sel(1ull << r, be.bcast(S, 0), L);
                S = be.shr(S, 1);
                pp = ppn;
            }
            S = be.sel(lanes(p), S, be.zero());
            C = be.sel(lanes(p), C, be.zero());
            return;
        }
        for (int r = 0; r < w; r++) {
            const int na = (M - r) < w ? (M - r) : w;
            const uint64_t act = lanes(na);
            W bb = be.bcast(b, r);
            W pp = be.AND(a, bb, act);
            uint64_t inv = (r == w - 1) ? lanes(w - 1) : (1ull << (w - 1));
            pp = be.NOTm(pp, inv & act);
            if (r == 0) {
                S = pp;
            } else {
                W t = be.AND(be.XOR(S, pp), be.XOR(C, pp), act);
                W Sn = be.XOR(be.XOR(S, C), pp);
                C = be.XOR(t, pp);
                S = Sn;
            }
            L = be.sel(1ull << r, be.bcast(S, 0), L);
            S = be.shr(S, 1);
        }
        S = be.sel(lanes(p), S, be.zero());
        C = be.sel(lanes(p), C, be.zero());
    }
    // X + Y = wrap_w((a*b) >> p)
    static GC_HD void mul_xy(B &be, W a, W b, int w, int p, W &X, W &Y) {
        W L, S, C;
        mul_core(be, a, b, w, p, L, S, C);
        const uint64_t act = lanes(w);
        X = be.XOR(be.sel(act, be.shr(L, p), be.zero()), be.sel(act, be.shl(S, w - p), be.zero()));
        Y = be.sel(act, be.shl(C, w - p), be.zero());
    }
    static GC_HD W mul(B &be, W a, W b, int w, int p) {
        W X, Y;
        mul_xy(be, a, b, w, p, X, Y);
        return add(be, X, Y, w);
    }
    // (AS, AC) += wrap_w((a*b) >> p), carry-save, mod 2^w
    static GC_HD void mac(B &be, W &AS, W &AC, W a, W b, int w, int p) {
        W X, Y;
        mul_xy(be, a, b, w, p, X, Y);
        csa(be, AS, AC, X, w);
        csa(be, AS, AC, Y, w);
    }

    // ---- two independent 32-bit products per wavefront (lanes 0..31 and 32..63): the 32-bit
    // build's multiply-accumulate would otherwise leave half of every wave idle.  Same circuit
    // as mul_core / mul_xy / csa with w = 32, every lane move confined to its 32-lane segment.
    static GC_HD uint64_t m2(uint64_t m32) { return (m32 & 0xffffffffull) | (m32 << 32); }
    static GC_HD W shr2(B &be, W x, int k) { return be.sel(m2(lanes(32 - k)), be.shr(x, k), be.zero()); }
    static GC_HD W shl2(B &be, W x, int k) { return be.sel(m2(lanes(32) & ~lanes(k)), be.shl(x, k), be.zero()); }
    static GC_HD void mul_core2(B &be, W a, W b, int p, W &L, W &S, W &C) {
        const int w = 32, M = w + p;
        S = be.zero();
        L = be.zero();
        C = (p >= 1) ? be.konst(m2(1ull << (w - 1))) : be.zero();
        for (int r = 0; r < w; r++) {
            const int na = (M - r) < w ? (M - r) : w;
            const uint64_t act = m2(lanes(na));
            W bb = be.bcast2(b, r);
            W pp = be.AND(a, bb, act);
            uint64_t inv = m2((r == w - 1) ? lanes(w - 1) : (1ull << (w - 1)));
            pp = be.NOTm(pp, inv & act);
            if (r == 0) {
                S = pp;
            } else {
                W t = be.AND(be.XOR(S, pp), be.XOR(C, pp), act);
                W Sn = be.XOR(be.XOR(S, C), pp);
                C = be.XOR(t, pp);
                S = Sn;
            }
            L = be.sel(m2(1ull << r), be.bcast2(S, 0), L);
            S = shr2(be, S, 1);
        }
        S = be.sel(m2(lanes(p)), S, be.zero());
        C = be.sel(m2(lanes(p)), C, be.zero());
    }
    // both segments: (AS, AC) += wrap_32((a*b) >> p), carry-save
    static GC_HD void mac2(B &be, W &AS, W &AC, W a, W b, int p) {
        W L, S, C;
        mul_core2(be, a, b, p, L, S, C);
        W X = be.XOR(shr2(be, L, p), shl2(be, S, 32 - p));
        W Y = shl2(be, C, 32 - p);
        const uint64_t all = ~0ull;
        W t = be.AND(be.XOR(AS, X), be.XOR(AC, X), all);
        W carry = be.XOR(t, X);
        AS = be.XOR(be.XOR(AS, AC), X);
        AC = shl2(be, carry, 1);
        t = be.AND(be.XOR(AS, Y), be.XOR(AC, Y), all);
        carry = be.XOR(t, Y);
        AS = be.XOR(be.XOR(AS, AC), Y);
        AC = shl2(be, carry, 1);
    }

    // ---- Karatsuba multiply-accumulate, w = 64 (the matrix-vector products of cgd.oc:119-125: 96 % of the gates of
    // a d = 500 solve).  With the operands read as unsigned words A = A1 2^32 + A0, B = B1 2^32 + B0,
    //     a b = A B - 2^64 (sa B + sb A)                       (mod 2^128; sa, sb: the sign bits)
    //     A B = Lo + 2^32 (H + Lo - DA DB) + 2^64 H,           Lo = A0 B0, H = A1 B1, DA = A1 - A0, DB = B1 - B0
    // so one product costs three unsigned 32 x 32 arrays -- Lo, H and DQ = |DA| |DB| -- instead of one 64 x 64 array.
    // A 32 x 32 array fills half a wave: TWO products run side by side (product k in lanes 0..31, product k' in lanes
    // 32..63 of every packed word), 3 x 63 gate steps per pair.  |DA| and the sign of DA do not depend on the other
    // operand: they are computed once per word (hdiff: once per solve for the matrix, once per iteration for the vector)
    // and stored `delta` words above the operand.  The sub-products arrive as L + 2^32 (S + C) (low half resolved,
    // high half carry-save); everything is then added in carry-save form:
    //     LOW  (bits 32..63, both products packed): S0 + C0 + L0 + L1 +- L2                    3 steps per pair
    //     the carry into bit p out of the dropped bits 32..p-1 (only L0 lives below bit 32):    1 + ceil(log2(p-32)) per pair
    //          only the carry OUT is wanted, so the (generate, propagate) pairs are combined in a tree anchored at the
    //          top lane, both ANDs of a node in ONE gate step (the second in the lane of the node just consumed)
    //     HIGH (bits 64..64+p-1, per product): L1|S1, S0|C1, C0, S1, C1, +-S2, +-C2 and the two sign corrections
    //          -(sa & B), -(sb & A) (2 steps to form them)                                      9 steps per product
    //     (AS, AC) += bits p..p+63                                                              2 steps per product
    // Negated operands enter complemented over a lane mask that reaches the top of the range, +1 at their lowest lane:
    // a CSA step has a free slot in lane 0 of its shifted carry word, and there are exactly as many slots as corrections
    // (LOW: n; HIGH: the three carries out of LOW, n, n, ~n, 1).  220 steps per pair at p = 56 against 2 x 129.
    static GC_HD W half_lo(B &be, W x, int h) { return h ? be.shr(x, 32) : be.sel(lanes(32), x, be.zero()); }
    static GC_HD W half_hi(B &be, W x, int h) { return h ? be.sel(~lanes(32), x, be.zero()) : be.shl(x, 32); }
    // two independent unsigned 32 x 32 products per wave: value = L + 2^32 (S + C) in each half
    static GC_HD void umul32x2(B &be, W a, W b, W &L, W &S, W &C) {
        const uint64_t all = ~0ull;
        S = be.zero(); L = be.zero(); C = be.zero();
        if (B::kPairSteps) {
            // the partial product of row r + 1 does not depend on row r: it issues with the carry-save AND of row r (same
            // gate-step numbers as the loop below: pp0, pp1, csa1, pp2, csa2, ...)
            W pp = be.AND(a, be.bcast2(b, 0), all);
            for (int r = 0; r < 32; r++) {
                W ppn = be.zero();
                if (r == 0) {
                    S = pp;
                    ppn = be.AND(a, be.bcast2(b, 1), all);
                } else {
                    W t;
                    if (r + 1 < 32) be.AND2(be.XOR(S, pp), be.XOR(C, pp), all, a, be.bcast2(b, r + 1), all, t, ppn);
                    else t = be.AND(be.XOR(S, pp), be.XOR(C, pp), all);
                    W Sn = be.XOR(be.XOR(S, C), pp);
                    C = be.XOR(t, pp);
                    S = Sn;
                }
                L = be.sel(m2(1ull << r), be.bcast2(S, 0), L);
                S = shr2(be, S, 1);
                pp = ppn;
            }
            return;
        }
        for (int r = 0; r < 32; r++) {
            W pp = be.AND(a, be.bcast2(b, r), all);
            if (r == 0) {
                S = pp;
            } else {
                W t = be.AND(be.XOR(S, pp), be.XOR(C, pp), all);
                W Sn = be.XOR(be.XOR(S, C), pp);
                C = be.XOR(t, pp);
                S = Sn;
            }
            L = be.sel(m2(1ull << r), be.bcast2(S, 0), L);
            S = shr2(be, S, 1);
        }
    }
    // |A1 - A0| in lanes 0..31, [A1 < A0] in lane 32
    static GC_HD W hdiff(B &be, W a) {
        W x = be.shr(a, 32), y = be.sel(lanes(32), a, be.zero()), ge;
        W d = sub(be, x, y, 32, &ge);
        W neg = be.NOTm(ge, ~0ull);
        W m = condneg(be, d, neg, 32);
        return be.XOR(be.sel(lanes(32), m, be.zero()), be.sel(1ull << 32, neg, be.zero()));
    }
    // (AS, AC) += wrap_64((a b) >> p) + wrap_64((a' b') >> p), carry-save.  a0 / a1, b0 / b1: the operand words of the two
    // products, da0 .. db1: the words holding their hdiff.  A single product is paired with the constant-zero word (id 0,
    // whose hdiff is word 0 as well).  The three arrays run as ONE loop -- a single copy of the two inlined gate bodies in
    // a GPU kernel -- whose results go to a small array indexed by the loop counter: on the GPU that array lives in
    // scratch memory, so the nine sub-product words do not occupy registers while the next array is hashed.  Operands are
    // loaded where they are used (load2s: lanes 32 half .. 32 half + 31 of two words, packed).
    static GC_HD void mack2(B &be, W &AS, W &AC, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t da0, uint32_t da1,
                            uint32_t db0, uint32_t db1, int p) {
        const uint64_t all = ~0ull;
        W sub[9];                                          // (L, S, C) of Lo = A0 B0, H = A1 B1, DQ = |DA| |DB|
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int t = 0; t < 3; t++) {
            const bool dq = t == 2;
            W L, S, C;
            umul32x2(be, be.load2s(dq ? da0 : a0, dq ? da1 : a1, t == 1), be.load2s(dq ? db0 : b0, dq ? db1 : b1, t == 1), L, S, C);
            sub[3 * t] = L; sub[3 * t + 1] = S; sub[3 * t + 2] = C;
        }
        // DA DB = (-1)^(sA ^ sB) DQ enters Mid with a minus sign: n = 1 where DQ is subtracted
        W n = be.NOTm(be.bcast2(be.XOR(be.load2s(da0, da1, 1), be.load2s(db0, db1, 1)), 0), all);
        // The recombination is written as two loops with ONE gate each (`switch` on the step number around it): a GPU
        // kernel then holds one inlined copy of the gate body per loop instead of one per carry-save step.
        // ---- phase A, both products packed (lane j of a half = bit 32 + j): LOW = S0 + C0 + L1 + L0 +- L2 in three
        // carry-save steps (carries out of bit 63 -> o1..o3), then the carry out of its lanes [0, p - 32) (carry2's tree)
        W SL = sub[1], CL = sub[2], o1 = be.zero(), o2 = be.zero(), o3 = be.zero(), cp = be.zero();
        {
            const int nb = p - 32;
            int levels = 0;
            for (int dist = 1; dist < nb; dist <<= 1) levels++;
            const int nsteps = 3 + (nb > 0 ? 1 + levels : 0);
            W X = be.zero(), G = be.zero(), Pg = be.zero();
            int dist = 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int i = 0; i < nsteps; i++) {
                W ga, gb;
                uint64_t act = all, mA = 0, mB = 0;
                if (i < 3) {
                    X = i == 0 ? sub[3] : (i == 1 ? sub[0] : be.XOR(sub[6], n));
                    ga = be.XOR(SL, X); gb = be.XOR(CL, X);
                } else if (i == 3) {
                    ga = SL; gb = CL; act = m2(lanes(nb));
                } else {
                    for (int l = nb - 1; l - dist >= 0; l -= 2 * dist) { mA |= 1ull << l; mB |= 1ull << (l - dist); }
                    ga = be.sel(m2(mA), Pg, shr2(be, Pg, dist));
                    gb = be.sel(m2(mA), shl2(be, G, dist), Pg);
                    act = m2(mA | mB);
                }
                W t = be.AND(ga, gb, act);
                if (i < 3) {
                    W carry = be.XOR(t, X);
                    SL = be.XOR(be.XOR(SL, CL), X);
                    W co = shr2(be, carry, 31);
                    if (i == 0) o1 = co; else if (i == 1) o2 = co; else o3 = co;
                    CL = shl2(be, carry, 1);
                    if (i == 2) CL = be.XOR(CL, be.sel(m2(1ull), n, be.zero()));
                } else if (i == 3) {
                    G = t; Pg = be.XOR(SL, CL);
                } else {
                    G = be.XOR(G, be.sel(m2(mA), t, be.zero()));
                    Pg = be.sel(m2(mA), shl2(be, t, dist), Pg);
                    dist <<= 1;
                }
            }
            if (nb > 0) cp = be.sel(m2(1ull), shr2(be, G, nb - 1), be.zero());
        }
        // ---- phase B, per product: HIGH (lane i = bit 64 + i, bits below 64 + p matter) = L1|S1 + S0|C1 + C0 + S1 + C1 +- S2
        // +- C2 - (sa & B) - (sb & A): steps 0..8 (5 and 7 form the sign corrections); steps 9, 10 add bits p..p+63 of
        // the product to the accumulator, which takes the place of (SS, CC) for those two steps
        const int nh = p;
        const uint64_t ah = lanes(nh);
        for (int h = 0; h < 2; h++) {
            W nn = be.bcast(n, 32 * h);
            W SS = be.XOR(half_lo(be, sub[3], h), half_hi(be, sub[4], h));
            W CC = be.XOR(half_lo(be, sub[1], h), half_hi(be, sub[5], h));
            W aw = be.load(h ? a1 : a0), bw = be.load(h ? b1 : b0);
            W X = be.zero(), Yacc = be.zero(), cin = be.zero();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int i = (nh > 0 ? 0 : 9); i < 11; i++) {
                W ga, gb;
                uint64_t act = ah;
                const bool corr = i == 5 || i == 7;
                if (corr) {
                    ga = i == 5 ? bw : aw;
                    gb = be.bcast(i == 5 ? aw : bw, 63);
                } else {
                    if (i < 5) {
                        const int k = i == 0 ? 2 : (i == 1 ? 4 : (i == 2 ? 5 : (i == 3 ? 7 : 8)));
                        X = half_lo(be, sub[k], h);
                        if (i >= 3) X = be.sel(ah, be.XOR(X, nn), be.zero());
                        cin = i == 0 ? half_lo(be, o1, h) : (i == 1 ? half_lo(be, o2, h) : (i == 2 ? half_lo(be, o3, h) : nn));
                    } else if (i == 6) {
                        cin = be.NOTm(nn, 1ull);                   // X: the correction formed in step 5
                    } else if (i == 8) {
                        cin = be.konst(1ull);
                    } else if (i == 9) {
                        SS = be.sel(ah, SS, be.zero());
                        CC = be.sel(ah, CC, be.zero());
                        W LS = be.XOR(half_lo(be, sub[0], h), half_hi(be, SL, h)), LC = half_hi(be, CL, h);
                        X = be.XOR(be.shr(LS, p), be.shl(SS, 64 - p));
                        Yacc = be.XOR(be.shr(LC, p), be.shl(CC, 64 - p));
                        SS = AS; CC = AC;
                        cin = half_lo(be, cp, h);
                        act = all;
                    } else {                                        // i == 10
                        X = Yacc;
                        cin = be.zero();
                        act = all;
                    }
                    ga = be.XOR(SS, X); gb = be.XOR(CC, X);
                }
                W t = be.AND(ga, gb, act);
                if (corr) {
                    X = be.NOTm(t, ah);
                } else {
                    W carry = be.XOR(t, be.sel(act, X, be.zero()));
                    SS = be.XOR(be.XOR(SS, CC), X);
                    CC = be.XOR(be.sel(act, be.shl(carry, 1), be.zero()), be.sel(1ull, cin, be.zero()));
                }
            }
            AS = SS; AC = CC;
        }
    }
    // the products of one OP_MACK record: pairs (2k, 2k + 1); a last single product is paired with the zero word
    static GC_HD void mack_rec(B &be, W &AS, W &AC, uint32_t a, uint32_t b, int32_t sa, int32_t sb, uint32_t cnt, uint32_t delta, int p) {
        for (uint32_t k = 0; k < cnt; k += 2) {
            const bool two = k + 1 < cnt;
            const uint32_t a0 = a + (int32_t)k * sa, b0 = b + (int32_t)k * sb, a1 = two ? a0 + sa : 0u, b1 = two ? b0 + sb : 0u;
            mack2(be, AS, AC, a0, a1, b0, b1, a0 + delta, two ? a1 + delta : 0u, b0 + delta, two ? b1 + delta : 0u, p);
        }
    }

    // ---- wide inner-product accumulator: sum of exact products mod 2^(w+p)
    struct IpAcc { W LS, LC, HS, HC; };
    static GC_HD void ip_zero(B &be, IpAcc &A) { A.LS = A.LC = A.HS = A.HC = be.zero(); }
    static GC_HD void ip_mac(B &be, IpAcc &A, W a, W b, int w, int p) {
        W L, S, C, co;
        mul_core(be, a, b, w, p, L, S, C);
        csa(be, A.LS, A.LC, L, w, be.zero(), &co);
        if (p > 0) {
            csa(be, A.HS, A.HC, S, p, co, (W *)0);
            csa(be, A.HS, A.HC, C, p);
        }
    }
    static GC_HD void ip_merge(B &be, IpAcc &A, const IpAcc &O, int w, int p) {
        W c1, c2;
        csa(be, A.LS, A.LC, O.LS, w, be.zero(), &c1);
        csa(be, A.LS, A.LC, O.LC, w, be.zero(), &c2);
        if (p > 0) {
            csa(be, A.HS, A.HC, O.HS, p, c1, (W *)0);
            csa(be, A.HS, A.HC, O.HC, p, c2, (W *)0);
        }
    }
    // wrap_w(sum >> p)
    static GC_HD W ip_final(B &be, const IpAcc &A, int w, int p) {
        W cl;
        W lo = add(be, A.LS, A.LC, w, be.zero(), &cl);
        W hi = (p > 0) ? add(be, A.HS, A.HC, p, be.sel(1ull, cl, be.zero()), (W *)0) : be.zero();
        const uint64_t act = lanes(w);
        return be.XOR(be.sel(act, be.shr(lo, p), be.zero()), be.sel(act, be.shl(hi, w - p), be.zero()));
    }

    // ---- non-restoring divider: wrap_w(tdiv(a << p, b)), truncation toward zero.
    // The partial remainder R in [-|b|, |b|) is a (w+1)-bit two's complement number: its low w bits
    // live in the lanes, its sign s is a separate broadcast wire (no spare lane at w = 64).  Step k:
    //     X = 2R + n_k;   R' = s ? X + |b| : X - |b|      (one Kogge-Stone add, the +-|b| select is an XOR)
    //     s' = R[w-1] ^ ~s ^ carry-out                    (bit w of the sum: free)
    //     q_k = ~s'
    // (the digits +-1 of the non-restoring recurrence, converted: Q = 2E + ~s_0 - 2^M with E = sum e_k 2^k,
    // e_k = ~s_{k+1}, e_{M-1} = 1, is bit for bit q_k = ~s_k).  7 dependent levels per quotient bit; the
    // restoring form needs an eighth for the remainder mux.  b == 0 is unspecified upstream
    // (fixed.oc:174-180); oracle and circuit define an all-ones magnitude (-1 for a >= 0, +1 for a < 0),
    // forced here by a zero detector on |b| (zcheck; not needed for a public non-zero divisor).
    // bounded: the caller guarantees |a| <= |b| (CGD's g_i / max_j |g_j|, cgd.oc:104-106, 153-155: the divisor IS the
    // largest of the magnitudes, formed by this circuit with the same abs and an unsigned compare at w = 64).  The quotient
    // is then at most 2^p, its bits above p are zero, and the first w - 1 steps of the recurrence -- which shift the top
    // w - 1 bits of |a| into the remainder while it stays below |b| -- are replaced by that state: R = |a| >> 1 (< |b| for
    // |b| >= 1), s = 0; p + 1 quotient bits instead of w + p.  |b| = 0 (then |a| = 0 too) is the zero detector's case as before.
    static GC_HD W div_mag(B &be, W ua, W ub, int w, int p, bool zcheck, bool bounded = false) {
        const int M = w + p;
        const uint64_t act = lanes(w);
        W R = be.zero(), Q = be.zero(), s = be.zero();
        if (bounded) R = be.sel(act, be.shr(ua, 1), be.zero());
        for (int k = bounded ? p : M - 1; k >= 0; k--) {
            W rtop = be.bcast(R, w - 1);
            W X = be.sel(act, be.shl(R, 1), be.zero());
            if (k >= p) X = be.XOR(X, be.sel(1ull, be.bcast(ua, k - p), be.zero()));
            W ns = be.NOTm(s, ~0ull);
            W Y = be.XOR(ub, be.sel(act, ns, be.zero()));
            W co;
            R = add(be, X, Y, w, be.sel(1ull, ns, be.zero()), &co);
            s = be.XOR(be.XOR(rtop, ns), co);
            if (k < w) Q = be.sel(1ull << k, be.NOTm(s, ~0ull), Q);
        }
        if (zcheck) {
            W nz = ub;                                   // OR of all lanes of |b| -> lane 0
            for (int dist = 32; dist >= 1; dist >>= 1) {
                if (dist >= w) continue;
                W sh = be.shr(nz, dist);
                W t = be.AND(nz, sh, lanes(dist));
                nz = be.XOR(be.XOR(nz, sh), t);
            }
            // Q | ~nz
            Q = be.NOTm(be.AND(be.NOTm(Q, act), be.bcast(nz, 0), act), act);
        }
        return Q;
    }
    // [a == b] in lane 0: OR of the lanes of a ^ b by halving (w - 1 AND gates in log2 w steps, as many gates as the
    // reference's comparison of two w-bit words), inverted
    static GC_HD W equal(B &be, W a, W b, int w) {
        W nz = be.sel(lanes(w), be.XOR(a, b), be.zero());
        for (int dist = 32; dist >= 1; dist >>= 1) {
            if (dist >= w) continue;
            W sh = be.shr(nz, dist);
            W t = be.AND(nz, sh, lanes(dist));
            nz = be.sel(lanes(dist), be.XOR(be.XOR(nz, sh), t), be.zero());
        }
        return be.sel(1ull, be.NOTm(nz, 1ull), be.zero());
    }
    static GC_HD W div(B &be, W a, W b, int w, int p, bool bounded = false) {
        W sa = be.bcast(a, w - 1), sb = be.bcast(b, w - 1);
        W ua = condneg(be, a, sa, w), ub = condneg(be, b, sb, w);
        W Q = div_mag(be, ua, ub, w, p, true, bounded);
        return condneg(be, Q, be.XOR(sa, sb), w);
    }
    // tdiv(a, c) for a public constant c > 0 (the normalizer of linear.oc:52-65; rounds 1-4 ran the divider above with a
    // constant divisor word: 882 gate steps against 61) by a multiplication with a public constant (Granlund & Montgomery, "Division by invariant
    // integers using multiplication", PLDI 1994, theorem 4.2 at precision w-1): with l = ceil(log2 c) and
    //     m = floor(2^(w-1+l) / c) + 1        (2^(w-1) < m < 2^w; idivc_magic in gc_program.h, on the host)
    // floor(n / c) = floor(m n / 2^(w-1+l)) for every 0 <= n <= 2^(w-1) and c >= 2: m c = 2^(w-1+l) + e with 1 <= e <= c,
    // so m n / 2^(w-1+l) = n/c + n e / (c 2^(w-1+l)) and the excess is at most 2^-l <= 1/c -- reached only by a power of
    // two c at n = 2^(w-1), where n/c is whole.  |a| <= 2^(w-1) covers INT_MIN.  c = 1 is a copy.
    // The product |a| m is a sum of popcount(m) shifted copies of |a| (m is public: which copies is wiring), kept in
    // carry-save form.  w = 64: the 128-bit accumulator is two words (low L, high H); the copy |a| << j lies in lanes
    // j..63 of L and lanes 0..j-1 of H -- a ROTATION of |a| -- so ONE 64-lane gate step adds it to both: lanes j.. work on
    // L, lanes ..j-1 on H, and the lanes of either word that the copy does not reach keep their sum and carry bits
    // (S + C is what counts, not a canonical split).  The carry out of lane j-1 of the H half lands in lane j of H's carry
    // word, which is still zero: copies are taken in ascending j and the running sum is below 2^(64+j').  w = 32: one
    // word holds the 64-bit product, a step works on the 32 lanes of its copy.
    // Steps: 2 conditional negates + popcount(m) - 2 + the final addition(s): about 58 at w = 64 (the divider: 882).
    static GC_HD W divc(B &be, W a, uint64_t m, int l, int w) {
        const uint64_t act = lanes(w);
        if (l == 0) return be.sel(act, a, be.zero());                      // c == 1
        W sa = be.bcast(a, w - 1);
        W ua = be.sel(act, condneg(be, a, sa, w), be.zero());
        W q;
        int seen = 0;
        if (w == 64) {
            W LS = be.zero(), LC = be.zero(), HS = be.zero(), HC = be.zero();
            for (int j = 0; j < 64; j++) {
                if (!((m >> j) & 1ull)) continue;
                W lo = be.shl(ua, j), hi = j ? be.shr(ua, 64 - j) : be.zero();
                if (seen == 0) { LS = lo; HS = hi; }
                else if (seen == 1) { LC = lo; HC = hi; }
                else {
                    const uint64_t hl = lanes(j);                          // lanes that work on H in this step
                    W X = be.sel(hl, hi, lo);
                    W S = be.sel(hl, HS, LS), C = be.sel(hl, HC, LC);
                    W t = be.AND(be.XOR(S, X), be.XOR(C, X), ~0ull);
                    W carry = be.XOR(t, X);                                // maj(S, C, X)
                    W sum = be.XOR(be.XOR(S, C), X);
                    W cup = be.shl(carry, 1);
                    LS = be.sel(hl, LS, sum);
                    HS = be.sel(hl, sum, HS);
                    LC = be.sel(hl, LC, be.sel(~lanes(j + 1), cup, be.zero()));                  // lane j: consumed
                    HC = be.sel(lanes(j + 1), be.sel(1ull, be.shr(carry, 63), cup), HC);         // lane 0: L's carry out
                }
                seen++;
            }
            W cl;
            (void)add(be, LS, LC, 64, be.zero(), &cl);
            W h = add(be, HS, HC, 64, be.sel(1ull, cl, be.zero()), (W *)0);
            q = be.shr(h, l - 1);
        } else {
            W S = be.zero(), C = be.zero();
            for (int j = 0; j < w; j++) {
                if (!((m >> j) & 1ull)) continue;
                W X = be.shl(ua, j);
                if (seen == 0) S = X;
                else if (seen == 1) C = X;
                else {
                    const uint64_t on = act << j;
                    W t = be.AND(be.XOR(S, X), be.XOR(C, X), on);
                    W carry = be.XOR(t, X);
                    W sum = be.XOR(be.XOR(S, C), X);
                    S = be.sel(on, sum, S);
                    C = be.sel(on << 1, be.shl(carry, 1), be.sel(on, be.zero(), C));
                }
                seen++;
            }
            W full = add(be, S, C, 2 * w, be.zero(), (W *)0);
            q = be.sel(act, be.shr(full, w + l - 1), be.zero());
        }
        return condneg(be, q, sa, w);
    }

    // ---- square root.  w = 64: floor(sqrt(a_u * 2^p)); w = 32: the explicit loop of
    // src/fixed.oc:228-240 on the low (32+p) bits, mirrored literally.
    // w = 64, w + p <= 124: non-restoring digit recurrence on an (n+2)-bit signed remainder (n root bits):
    //     R >= 0: R = 4R + v_i - (4Q + 1)     R < 0: R = 4R + v_i + (4Q + 3)     q_i = (R >= 0), Q = 2Q + q_i
    // one Kogge-Stone add per root bit (7 levels; the restoring form adds a mux level).
    // w + p > 124 (precision 61..63, allowed by src/cmd/linreg.c:85-88): the remainder no longer fits the
    // 64 lanes; a restoring recurrence on a two-word remainder (low 64 lanes + 1 or 2 high lanes).
    static GC_HD W vsqrt(B &be, W a, int w, int p) {
        const int M = w + p;
        if (w == 32) {
            const uint64_t mk = lanes(M);
            W x = be.sel(mk, be.shl(be.sel(lanes(32), a, be.zero()), p), be.zero());
            W r = be.zero();
            for (int t = M - 2; t >= 0; t -= 2) {
                W re = be.NOTm(r, 1ull << t);            // r + e == r | e
                W co;
                W xs = sub(be, x, re, M, &co);           // co = (x >= r + e)
                x = mux(be, co, xs, x, M);
                r = be.sel(1ull << t, co, be.shr(r, 1)); // (r >> 1) + (co ? e : 0)
            }
            return be.sel(lanes(32), r, be.zero());
        }
        const int n = (M + 1) / 2;                       // root bits
        const int rb = n + 2;                            // remainder width
        W Q = be.zero();
        if (rb <= 64) {
            const uint64_t act = lanes(rb);
            W R = be.zero();
            for (int i = n - 1; i >= 0; i--) {
                W s = be.bcast(R, rb - 1);               // sign of the remainder before the step
                W ns = be.NOTm(s, ~0ull);
                W X = be.sel(act, be.shl(R, 2), be.zero());
                int hi = 2 * i + 1 - p, lo = 2 * i - p;
                if (hi >= 0 && hi < w) X = be.XOR(X, be.sel(2ull, be.bcast(a, hi), be.zero()));
                if (lo >= 0 && lo < w) X = be.XOR(X, be.sel(1ull, be.bcast(a, lo), be.zero()));
                // s = 0: - (4Q + 1) = ~(4Q + 1) + 1;   s = 1: + (4Q + 3)
                W Y = be.XOR(be.NOTm(be.sel(act, be.shl(Q, 2), be.zero()), 1ull), be.sel(2ull, s, be.zero()));
                Y = be.XOR(Y, be.sel(act, ns, be.zero()));
                R = add(be, X, Y, rb, be.sel(1ull, ns, be.zero()), (W *)0);
                W q = be.NOTm(be.bcast(R, rb - 1), ~0ull);
                Q = be.XOR(be.sel(act, be.shl(Q, 1), be.zero()), be.sel(1ull, q, be.zero()));
            }
            return be.sel(lanes(w), Q, be.zero());
        }
        const int nh = rb - 64;                          // 1 or 2 high lanes
        W Rl = be.zero(), Rh = be.zero();
        for (int i = n - 1; i >= 0; i--) {
            Rh = be.sel(lanes(nh), be.XOR(be.shl(Rh, 2), be.shr(Rl, 62)), be.zero());
            Rl = be.shl(Rl, 2);
            int hi = 2 * i + 1 - p, lo = 2 * i - p;
            if (hi >= 0 && hi < w) Rl = be.XOR(Rl, be.sel(2ull, be.bcast(a, hi), be.zero()));
            if (lo >= 0 && lo < w) Rl = be.XOR(Rl, be.sel(1ull, be.bcast(a, lo), be.zero()));
            W Tl = be.NOTm(be.shl(Q, 2), 1ull);          // 4Q + 1, low 64 bits
            W Th = be.sel(lanes(nh), be.shr(Q, 62), be.zero());
            W c1, c2;
            W Dl = sub(be, Rl, Tl, 64, &c1);
            W Dh = add(be, Rh, be.NOTm(Th, lanes(nh)), nh, be.sel(1ull, c1, be.zero()), &c2);
            W ml, mh;
            be.AND2(c2, be.XOR(Dl, Rl), ~0ull, c2, be.XOR(Dh, Rh), lanes(nh), ml, mh);
            Rl = be.XOR(Rl, ml);
            Rh = be.XOR(Rh, mh);
            Q = be.XOR(be.shl(Q, 1), be.sel(1ull, c2, be.zero()));
        }
        return Q;
    }
};

// ---- host-side plaintext backend: circuit logic checks + step/gate counting
struct PlainBackend {
    typedef uint64_t W;
    static const bool kPairSteps = false;   // order of independent gate steps in mul_core (see there)
    uint64_t steps, gates;
    PlainBackend() : steps(0), gates(0) {}
    GC_HD W zero() const { return 0; }
    GC_HD W konst(uint64_t bits) const { return bits; }
    GC_HD W XOR(W a, W b) const { return a ^ b; }
    GC_HD W NOTm(W a, uint64_t m) const { return a ^ m; }
    GC_HD W AND(W a, W b, uint64_t act) {
        steps++;
        gates += (uint64_t)__builtin_popcountll(act);
        return a & b & act;
    }
    // two independent gate steps (numbered step, step+1) that a backend may run concurrently
    GC_HD void AND2(W a1, W b1, uint64_t act1, W a2, W b2, uint64_t act2, W &c1, W &c2) {
        c1 = AND(a1, b1, act1);
        c2 = AND(a2, b2, act2);
    }
    GC_HD W shl(W a, int k) const { return k >= 64 ? 0 : a << k; }
    GC_HD W shr(W a, int k) const { return k >= 64 ? 0 : a >> k; }
    GC_HD W bcast(W a, int lane) const { return ((a >> lane) & 1) ? ~0ull : 0ull; }
    // per 32-lane segment: lanes 0..31 <- lane r, lanes 32..63 <- lane 32 + r
    GC_HD W bcast2(W a, int r) const {
        return (((a >> r) & 1) ? 0xffffffffull : 0ull) | (((a >> (32 + r)) & 1) ? 0xffffffff00000000ull : 0ull);
    }
    GC_HD W sel(uint64_t m, W a, W b) const { return (a & m) | (b & ~m); }
    // every lane <- the top lane of the LOWER half of its block of 2^(k+1) lanes: lane (l & ~(2^(k+1) - 1)) | (2^k - 1)
    GC_HD W bblk(W a, int k) const {
        const int h = 1 << k;
        W r = 0;
        for (int base = 0; base < 64; base += 2 * h)
            if ((a >> (base + h - 1)) & 1ull) r |= (2 * h >= 64 ? ~0ull : (((1ull << (2 * h)) - 1) << base));
        return r;
    }
};

}  // namespace gc
