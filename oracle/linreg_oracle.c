/*
 * linreg_oracle.c -- CPU semantic oracle.  TEST INFRASTRUCTURE ONLY: see the
 * header.  Plain integers + IEEE doubles, no crypto.  Build with
 * -ffp-contract=off (the diagonal of A is a floating-point computation whose
 * rounding is part of bit-exactness, reference src/phase1.c:562-567).
 */
#include "linreg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef __int128 i128;

/* ------------------------------------------------------------------ scalars */

int64_t orc_wrap(int64_t v, int w) {
    return w == 32 ? (int64_t)(int32_t)(uint32_t)(uint64_t)v : v;
}

/* (fixed_t)(d * (1ll << p)): C truncation toward zero (src/fixed.c:3-5).
 * Out-of-range casts are undefined in C; the reference runs on x86-64 where
 * cvttsd2si returns the "integer indefinite" value, mirrored here. */
int64_t orc_double_to_fixed(double d, int p, int w) {
    double t = d * (double)(1ll << p);
    if (w == 32) {
        if (!(t > -2147483649.0 && t < 2147483648.0)) return (int64_t)INT32_MIN;
        return (int64_t)(int32_t)t;
    }
    if (!(t >= -9223372036854775808.0 && t < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)t;
}

/* ((double) f) / (1ll << p)  (src/fixed.c:7-9) */
double orc_fixed_to_double(int64_t f, int p) {
    return ((double)f) / (double)(1ll << p);
}

int64_t orc_add(int64_t a, int64_t b, int w) {
    return orc_wrap((int64_t)((uint64_t)a + (uint64_t)b), w);
}

int64_t orc_sub(int64_t a, int64_t b, int w) {
    return orc_wrap((int64_t)((uint64_t)a - (uint64_t)b), w);
}

/* obliv if(a < 0) *out = 0-a; else *out = a;  (src/fixed.oc:90-97) */
int64_t orc_abs(int64_t a, int w) {
    return a < 0 ? orc_sub(0, a, w) : a;
}

/* W=32: signed three-way compare; W=64: obig_cmp, unsigned ("comparison only
 * works for unsigned numbers", src/fixed.oh:11; src/fixed.oc:78-88). */
int orc_cmp(int64_t a, int64_t b, int w) {
    if (w == 32) return a < b ? -1 : (a > b ? 1 : 0);
    uint64_t ua = (uint64_t)a, ub = (uint64_t)b;
    return ua < ub ? -1 : (ua > ub ? 1 : 0);
}

/* wrap_W( (a*b) >> p ), product exact, arithmetic shift (src/fixed.oc:149-162) */
int64_t orc_mul(int64_t a, int64_t b, int p, int w) {
    if (w == 32) {
        int64_t prod = (int64_t)((uint64_t)a * (uint64_t)b); /* both fit 32 bits: exact */
        return orc_wrap(prod >> p, 32);
    }
    i128 prod = (i128)a * (i128)b;
    return (int64_t)(uint64_t)(u128)(prod >> p);
}

/* wrap_W( tdiv(a * 2^p, b) ), C truncation toward zero (src/fixed.oc:164-188).
 * b == 0 is unspecified in the reference (the zero check is commented out at
 * fixed.oc:174-180); this build's restoring divider yields an all-ones
 * magnitude, i.e. -1 for a >= 0 and +1 for a < 0, and the oracle says so. */
int64_t orc_div(int64_t a, int64_t b, int p, int w) {
    if (b == 0) return a < 0 ? 1 : -1;
    if (w == 32) {
        int64_t num = (int64_t)((uint64_t)a << p);
        return orc_wrap(num / b, 32);
    }
    i128 num = (i128)a * ((i128)1 << p);
    i128 q = num / (i128)b;
    return (int64_t)(uint64_t)(u128)q;
}

static uint64_t isqrt_u128(u128 v) {
    /* bit-by-bit floor square root */
    u128 r = 0, e = (u128)1 << 126;
    while (e > v) e >>= 2;
    while (e != 0) {
        if (v >= r + e) { v -= r + e; r = (r >> 1) + e; }
        else r >>= 1;
        e >>= 2;
    }
    return (uint64_t)r;
}

/* floor(sqrt(a * 2^p)).  W=32 follows the explicit loop at src/fixed.oc:228-240
 * on the low (32+p) bits; W=64 is obig_sqrt of the shifted value
 * (fixed.oc:243-246).  Negative inputs are unspecified in the reference; both
 * paths here read the operand's low W bits as an unsigned number. */
int64_t orc_sqrt(int64_t a, int p, int w) {
    if (w == 32) {
        uint64_t mask = (((uint64_t)1) << (32 + p)) - 1;
        uint64_t x = (((uint64_t)a) << p) & mask;
        uint64_t r = 0;
        for (uint64_t e = mask + 1; e != 0; e >>= 2) {
            if ((x & mask) >= ((r + e) & mask)) {
                x -= r + e;
                r = ((r >> 1) + e) & mask;
            } else {
                r = r >> 1;
            }
        }
        return orc_wrap((int64_t)r, 32);
    }
    u128 v = ((u128)(uint64_t)a) << p;
    return (int64_t)isqrt_u128(v);
}

/* wrap_W( (sum_i a_i*b_i) >> p ): no intermediate truncation, one shift
 * (src/fixed.oc:124-147).  W=64: exact sum (only bits p..p+63 matter, so the
 * sum is kept mod 2^128).  W=32: wrapping int64 accumulator, declared without
 * initialiser at fixed.oc:126 and treated as 0. */
int64_t orc_inner_product(const int64_t *a, const int64_t *b, size_t d, int p, int w) {
    if (w == 32) {
        uint64_t acc = 0;
        for (size_t i = 0; i < d; i++) acc += (uint64_t)a[i] * (uint64_t)b[i];
        return orc_wrap(((int64_t)acc) >> p, 32);
    }
    u128 acc = 0;
    for (size_t i = 0; i < d; i++) acc += (u128)((i128)a[i] * (i128)b[i]);
    return (int64_t)(uint64_t)(u128)(((i128)acc) >> p);
}

/* packed lower-triangle index (src/linear.c:11-16) */
size_t orc_idx(size_t i, size_t j) {
    if (j > i) { size_t t = i; i = j; j = t; }
    return (i * (i + 1)) / 2 + j;
}

/* ------------------------------------------------------------ quantisation */

/* normalizer = sqrt(pow(2,p1) * n); val /= normalizer; double_to_fixed(val, p1)
 * (src/phase1.c:473-476; src/linear.c:46-51, 82-87).  The cast goes through
 * the phase-2 type fixed_t even though the buffer is fixed_p1_t. */
void orc_quantize(const double *vals, size_t count, int p1, size_t n, int w2, int64_t *out) {
    double normalizer = sqrt(pow(2, p1) * (double)n);
    for (size_t k = 0; k < count; k++) {
        double v = vals[k];
        v /= normalizer;
        out[k] = orc_double_to_fixed(v, p1, w2);
    }
}

/* -------------------------------------------------------------- aggregation */

static uint64_t mask_w(int w) { return w == 32 ? 0xffffffffull : ~0ull; }

/* inner_product_local: unsigned W-bit wraparound MAC (src/phase1.c:14-20) */
static uint64_t ip_local(const int64_t *x, size_t sx, const int64_t *y, size_t sy, size_t n, int w) {
    uint64_t acc = 0;
    for (size_t k = 0; k < n; k++) acc += (uint64_t)x[k * sx] * (uint64_t)y[k * sy];
    return acc & mask_w(w);
}

/* diagonal special case, floating point (src/phase1.c:562-567 and 364-369):
 *   xy += pow(fixed_to_double_p1(x_k, p), 2) * pow(2, p);  k ascending
 *   share = double_to_fixed_p1(xy / normalizer2, p),  normalizer2 = d.
 * pow(v, 2) is restated as v*v (gcc expands pow(x, 2.0) to x*x at -O3). */
static uint64_t diag_share(const int64_t *x, size_t sx, size_t n, size_t d, int p, int w1) {
    double xy = 0;
    for (size_t k = 0; k < n; k++) {
        double v = orc_fixed_to_double(x[k * sx], p);
        xy += (v * v) * pow(2, p);
    }
    return (uint64_t)orc_double_to_fixed(xy / (double)d, p, w1) & mask_w(w1);
}

void orc_aggregate(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                   int p1, int w1, uint64_t *A, uint64_t *b) {
    for (size_t i = 0; i < d; i++) {
        for (size_t j = 0; j <= i; j++) {
            if (i == j) A[orc_idx(i, j)] = diag_share(Xq + i, d, n, d, p1, w1);
            else A[orc_idx(i, j)] = ip_local(Xq + i, d, Xq + j, d, n, w1);
        }
        b[i] = ip_local(Xq + i, d, yq, 1, n, w1);
    }
}

/* get_owner (src/phase1.c:25-33), DP index 0..P-1; row d (target) -> last DP */
static size_t owner_of(size_t row, size_t P, const size_t *start) {
    size_t party = 0;
    while (party + 1 < P && start[party + 1] <= row) party++;
    return party;
}

/* The TI's randomness arrives through `next`: every cross-party pair asks for its 2n+1 words (x, y, r
 * in this order, phase1.c:271-273), so a full-size run never holds the whole stream in memory. */
int orc_phase1_ti_shares_cb(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                            int p1, int w1, size_t P, const size_t *start,
                            orc_rnd_fn next, void *ctx, size_t *rnd_words,
                            uint64_t *shares_A, uint64_t *shares_b) {
    size_t T = d * (d + 1) / 2, used = 0;
    uint64_t m = mask_w(w1);
    memset(shares_A, 0, P * T * sizeof(uint64_t));
    memset(shares_b, 0, P * d * sizeof(uint64_t));
    uint64_t *bx = malloc(n * sizeof(uint64_t)), *ay = malloc(n * sizeof(uint64_t));
    uint64_t *rnd = malloc((2 * n + 1) * sizeof(uint64_t));
    if (!bx || !ay || !rnd) { free(bx); free(ay); free(rnd); return 1; }
    /* loop order of run_party / run_trusted_initializer (phase1.c:256-258, 534-545) */
    for (size_t i = 0; i <= d; i++) {
        const int64_t *row_i = i < d ? Xq + i : yq;
        size_t stride_i = i < d ? d : 1;
        for (size_t j = 0; j <= i && j < d; j++) {
            const int64_t *row_j = Xq + j;
            size_t oi = owner_of(i, P, start), oj = owner_of(j, P, start);
            uint64_t sa, sb = 0;
            int two = 0;
            if (i == j) {
                sa = diag_share(row_i, stride_i, n, d, p1, w1);
            } else if (oi == oj) {
                sa = ip_local(row_i, stride_i, row_j, d, n, w1);
            } else {
                /* TI: x, y, r in this order (phase1.c:271-273); a = owner(i) gets
                 * (y, <x,y>-r), b = owner(j) gets (x, r) (phase1.c:253-254, 277-284) */
                if (next(ctx, used, 2 * n + 1, rnd)) { free(bx); free(ay); free(rnd); return 2; }
                const uint64_t *x = rnd, *y = rnd + n;
                uint64_t r = rnd[2 * n] & m;
                used += 2 * n + 1;
                uint64_t xy = 0;
                for (size_t k = 0; k < n; k++) xy += (x[k] & m) * (y[k] & m);
                xy &= m;
                for (size_t k = 0; k < n; k++) {
                    bx[k] = ((uint64_t)row_j[k * d] + (x[k] & m)) & m;        /* b + x -> a (201-207) */
                    ay[k] = ((uint64_t)row_i[k * stride_i] - (y[k] & m)) & m; /* a - y -> b (186-191) */
                }
                /* a: <b+x, y> - (<x,y> - r) (194-196); b: <a-y, b> - r (220-222) */
                uint64_t s1 = 0, s2 = 0;
                for (size_t k = 0; k < n; k++) {
                    s1 += bx[k] * (y[k] & m);
                    s2 += ay[k] * (uint64_t)row_j[k * d];
                }
                sa = (s1 - ((xy - r) & m)) & m;
                sb = (s2 - r) & m;
                two = 1;
            }
            if (i < d) {
                shares_A[oi * T + orc_idx(i, j)] = sa;
                if (two) shares_A[oj * T + orc_idx(i, j)] = sb;
            } else {
                shares_b[oi * d + j] = sa;
                if (two) shares_b[oj * d + j] = sb;
            }
        }
    }
    free(bx); free(ay); free(rnd);
    *rnd_words = used;
    return 0;
}

struct rnd_array { const uint64_t *words; size_t cap; };
static int rnd_from_array(void *ctx, size_t first, size_t count, uint64_t *out) {
    const struct rnd_array *a = ctx;
    if (first + count > a->cap) return 1;
    memcpy(out, a->words + first, count * sizeof(uint64_t));
    return 0;
}
int orc_phase1_ti_shares(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                         int p1, int w1, size_t P, const size_t *start,
                         const uint64_t *rnd, size_t *rnd_words,
                         uint64_t *shares_A, uint64_t *shares_b) {
    struct rnd_array a = {rnd, *rnd_words};
    return orc_phase1_ti_shares_cb(Xq, yq, n, d, p1, w1, P, start, rnd_from_array, &a, rnd_words, shares_A, shares_b);
}

/* one Gilboa inner product (src/phase1.c:38-96): receiver holds a (choice
 * bits LSB first per word), sender holds b; per OT the sender's pair is
 * (s, 2^bit * b_k + s); sender share -sum(s), receiver share sum(chosen). */
static void gilboa(const int64_t *a, size_t sa, const int64_t *b, size_t sb, size_t n, int w,
                   const uint64_t *s, uint64_t *share_sender, uint64_t *share_recver) {
    uint64_t m = mask_w(w), ss = 0, sr = 0;
    for (size_t k = 0; k < n; k++) {
        uint64_t ak = (uint64_t)a[k * sa] & m, bk = (uint64_t)b[k * sb] & m;
        for (int bit = 0; bit < w; bit++) {
            uint64_t s_i = s[k * (size_t)w + (size_t)bit] & m;
            uint64_t t = ((((uint64_t)1) << bit) * bk + s_i) & m;
            ss -= s_i;
            sr += ((ak >> bit) & 1) ? t : s_i;
        }
    }
    *share_sender = ss & m;
    *share_recver = sr & m;
}

int orc_phase1_ot_shares(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                         int p1, int w1, size_t P, const size_t *start,
                         const uint64_t *rnd, size_t *rnd_words,
                         uint64_t *shares_A, uint64_t *shares_b) {
    size_t T = d * (d + 1) / 2, used = 0, cap = *rnd_words, per = n * (size_t)w1;
    memset(shares_A, 0, P * T * sizeof(uint64_t));
    memset(shares_b, 0, P * d * sizeof(uint64_t));
#define COL_END(k) ((k) + 1 < P ? start[(k) + 1] : d)
    /* local blocks (run_party_ot_thread self branch, phase1.c:359-384) */
    for (size_t me = 0; me < P; me++) {
        for (size_t i = start[me]; i < COL_END(me); i++) {
            for (size_t j = start[me]; j <= i; j++) {
                shares_A[me * T + orc_idx(i, j)] =
                    i == j ? diag_share(Xq + i, d, n, d, p1, w1)
                           : ip_local(Xq + i, d, Xq + j, d, n, w1);
            }
            if (me == P - 1) shares_b[me * d + i] = ip_local(Xq + i, d, yq, 1, n, w1);
        }
    }
    /* cross blocks: sender iff ((me%2 == peer%2) == (me < peer)) (phase1.c:392) */
    for (size_t lo = 0; lo < P; lo++) {
        for (size_t hi = lo + 1; hi < P; hi++) {
            size_t pi = ((lo % 2) == (hi % 2)) ? lo : hi;  /* sender, party_i */
            size_t pj = pi == lo ? hi : lo;               /* receiver, party_j */
            uint64_t ss, sr;
            for (size_t i = start[pi]; i < COL_END(pi); i++) {
                for (size_t j = start[pj]; j < COL_END(pj); j++) {
                    if (used + per > cap) return 2;
                    gilboa(Xq + j, d, Xq + i, d, n, w1, rnd + used, &ss, &sr);
                    used += per;
                    shares_A[pi * T + orc_idx(i, j)] = ss;
                    shares_A[pj * T + orc_idx(i, j)] = sr;
                }
                if (pj == P - 1) { /* (i, target): phase1.c:425-435 */
                    if (used + per > cap) return 2;
                    gilboa(yq, 1, Xq + i, d, n, w1, rnd + used, &ss, &sr);
                    used += per;
                    shares_b[pi * d + i] = ss;
                    shares_b[pj * d + i] = sr;
                }
            }
            if (pi == P - 1) { /* party i owns the target (phase1.c:437-449) */
                for (size_t j = start[pj]; j < COL_END(pj); j++) {
                    if (used + per > cap) return 2;
                    gilboa(Xq + j, d, yq, 1, n, w1, rnd + used, &ss, &sr);
                    used += per;
                    shares_b[pi * d + j] = ss;
                    shares_b[pj * d + j] = sr;
                }
            }
        }
    }
#undef COL_END
    *rnd_words = used;
    return 0;
}

/* (ufixed_t)(((fixed_p1_t) share) >> (precision - precision_p2))  (phase1.c:609-638) */
void orc_convert_shares(const uint64_t *in, size_t count, int p1, int p2, int w1, int w2, uint64_t *out) {
    for (size_t k = 0; k < count; k++) {
        if (w1 == 64 && w2 == 32) out[k] = (uint64_t)(uint32_t)(uint64_t)(((int64_t)in[k]) >> (p1 - p2));
        else out[k] = in[k] & mask_w(w2);
    }
}

/* ------------------------------------------------------------ circuit input */

void orc_sum_shares(const uint64_t *shares, size_t P, size_t count, int w2, int64_t *out) {
    for (size_t k = 0; k < count; k++) {
        uint64_t s = 0;
        for (size_t q = 0; q < P; q++) s += shares[q * count + k];
        out[k] = orc_wrap((int64_t)s, w2);
    }
}

/* linear.oc:52-65: diag += double_to_fixed(lambda, p); off-diag and b:
 * ofixed_export(.) / (fixed_t) normalizer, C signed division, normalizer = d */
void orc_circuit_input(int64_t *a, int64_t *b, size_t d, double lambda, int p, int w) {
    int64_t lam = orc_double_to_fixed(lambda, p, w);
    for (size_t i = 0; i < d; i++) {
        for (size_t j = 0; j <= i; j++) {
            size_t ij = orc_idx(i, j);
            if (i == j) a[ij] = orc_add(a[ij], lam, w);
            else a[ij] = orc_wrap(a[ij] / (int64_t)d, w);
        }
    }
    for (size_t i = 0; i < d; i++) b[i] = orc_wrap(b[i] / (int64_t)d, w);
}

/* ------------------------------------------------------------------ solvers */

void orc_cgd(const int64_t *a, const int64_t *b, size_t d, int p, int w, int iters,
             int64_t *beta, int64_t *trace) {
    int64_t *x = calloc(d, 8), *pv = calloc(d, 8), *g = calloc(d, 8);
    int64_t *gscl = calloc(d, 8), *pA = calloc(d, 8);
    int64_t ng = 0, q = 0, eta = 0, gamma = 0, gAp, gp, t;
    /* cgd.oc:96-106 */
    for (size_t i = 0; i < d; i++) {
        g[i] = orc_sub(g[i], b[i], w);
        t = orc_abs(g[i], w);
        if (orc_cmp(t, ng, w) > 0) ng = t;
    }
    for (size_t i = 0; i < d; i++) pv[i] = orc_div(g[i], ng, p, w);
    for (int it = 0; it < iters; it++) {
        /* cgd.oc:119-125: each product shifted + wrapped, then wrap-added in j order */
        for (size_t i = 0; i < d; i++) {
            pA[i] = 0;
            for (size_t j = 0; j < d; j++)
                pA[i] = orc_add(pA[i], orc_mul(a[orc_idx(i, j)], pv[j], p, w), w);
        }
        q = orc_inner_product(pA, pv, d, p, w);          /* :128 */
        gp = orc_inner_product(g, pv, d, p, w);          /* :130 */
        eta = orc_div(gp, q, p, w);                      /* :133 */
        ng = 0;                                          /* :140 */
        for (size_t i = 0; i < d; i++) {                 /* :141-150 */
            x[i] = orc_sub(x[i], orc_mul(pv[i], eta, p, w), w);
            g[i] = orc_sub(g[i], orc_mul(eta, pA[i], p, w), w);
            t = orc_abs(g[i], w);
            if (orc_cmp(t, ng, w) > 0) ng = t;
        }
        for (size_t i = 0; i < d; i++) gscl[i] = orc_div(g[i], ng, p, w); /* :153-155 */
        gAp = orc_inner_product(pA, gscl, d, p, w);      /* :157 */
        gamma = orc_div(gAp, q, p, w);                   /* :159 */
        for (size_t i = 0; i < d; i++)                   /* :162-165 */
            pv[i] = orc_sub(gscl[i], orc_mul(pv[i], gamma, p, w), w);
        if (trace) {                                     /* reveals at :167-189 */
            int64_t *row = trace + (size_t)it * (d + 4);
            memcpy(row, x, d * 8);
            row[d] = gamma; row[d + 1] = eta; row[d + 2] = q; row[d + 3] = ng;
        }
    }
    memcpy(beta, x, d * 8);                              /* :201-208 */
    free(x); free(pv); free(g); free(gscl); free(pA);
}

void orc_cholesky(const int64_t *a_in, const int64_t *b_in, size_t d, int p, int w, int64_t *beta) {
    size_t T = d * (d + 1) / 2;
    int64_t *a = malloc(T * 8), *b = malloc(d * 8), *y = calloc(d, 8);
    memcpy(a, a_in, T * 8); memcpy(b, b_in, d * 8);
    memset(beta, 0, d * 8);
    /* cholesky.oc:51-65 */
    for (size_t j = 0; j < d; j++) {
        for (size_t k = 0; k < j; k++)
            for (size_t i = j; i < d; i++)
                a[orc_idx(i, j)] = orc_sub(a[orc_idx(i, j)],
                                           orc_mul(a[orc_idx(i, k)], a[orc_idx(j, k)], p, w), w);
        a[orc_idx(j, j)] = orc_sqrt(a[orc_idx(j, j)], p, w);
        for (size_t k = j + 1; k < d; k++)
            a[orc_idx(k, j)] = orc_div(a[orc_idx(k, j)], a[orc_idx(j, j)], p, w);
    }
    /* :68-76 */
    for (size_t i = 0; i < d; i++) {
        for (size_t j = 0; j < i; j++)
            b[i] = orc_sub(b[i], orc_mul(a[orc_idx(i, j)], y[j], p, w), w);
        y[i] = orc_div(b[i], a[orc_idx(i, i)], p, w);
    }
    /* :79-87 */
    for (size_t ii = d; ii-- > 0;) {
        for (size_t j = d; j-- > ii + 1;)
            y[ii] = orc_sub(y[ii], orc_mul(a[orc_idx(j, ii)], beta[j], p, w), w);
        beta[ii] = orc_div(y[ii], a[orc_idx(ii, ii)], p, w);
    }
    free(a); free(b); free(y);
}

void orc_ldlt(const int64_t *a_in, const int64_t *b_in, size_t d, int p, int w, int64_t *beta) {
    size_t T = d * (d + 1) / 2;
    int64_t *a = malloc(T * 8), *b = malloc(d * 8);
    memcpy(a, a_in, T * 8); memcpy(b, b_in, d * 8);
    /* ldlt.oc:50-64 */
    for (size_t j = 0; j < d; j++) {
        for (size_t k = 0; k < j; k++) {
            int64_t a_jk_kk = orc_mul(a[orc_idx(j, k)], a[orc_idx(k, k)], p, w);
            for (size_t i = j; i < d; i++)
                a[orc_idx(i, j)] = orc_sub(a[orc_idx(i, j)],
                                           orc_mul(a[orc_idx(i, k)], a_jk_kk, p, w), w);
        }
        for (size_t k = j + 1; k < d; k++)
            a[orc_idx(k, j)] = orc_div(a[orc_idx(k, j)], a[orc_idx(j, j)], p, w);
    }
    /* :67-73 */
    for (size_t i = 0; i < d; i++)
        for (size_t j = 0; j < i; j++)
            b[i] = orc_sub(b[i], orc_mul(a[orc_idx(i, j)], b[j], p, w), w);
    /* :76-79 */
    for (size_t i = 0; i < d; i++) b[i] = orc_div(b[i], a[orc_idx(i, i)], p, w);
    /* :82-90 */
    for (size_t ii = d; ii-- > 0;) {
        for (size_t j = d; j-- > ii + 1;)
            b[ii] = orc_sub(b[ii], orc_mul(a[orc_idx(j, ii)], b[j], p, w), w);
        beta[ii] = b[ii];
    }
    free(a); free(b);
}

/* --------------------------------------------------------------- input file */

int orc_read_input(const char *path, orc_input *in) {
    memset(in, 0, sizeof(*in));
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    char ep[512];
    size_t n2, d2;
    /* config.c:24-44 */
    if (fscanf(f, "%zu %zu %zu", &in->n, &in->d, &in->P) != 3) goto bad;
    in->start = calloc(in->P, sizeof(size_t));
    for (size_t i = 0; i < in->P + 2; i++) {
        if (fscanf(f, "%511s", ep) != 1) goto bad;
        if (i >= 2 && fscanf(f, "%zu", &in->start[i - 2]) != 1) goto bad;
    }
    /* linear.c:33-36 */
    if (fscanf(f, "%zu %zu", &n2, &d2) != 2 || n2 != in->n || d2 != in->d) goto bad;
    in->X = malloc(in->n * in->d * sizeof(double));
    for (size_t k = 0; k < in->n * in->d; k++)
        if (fscanf(f, "%lf", &in->X[k]) != 1) goto bad;
    /* linear.c:73 */
    if (fscanf(f, "%zu", &n2) != 1 || n2 != in->n) goto bad;
    in->y = malloc(in->n * sizeof(double));
    for (size_t k = 0; k < in->n; k++)
        if (fscanf(f, "%lf", &in->y[k]) != 1) goto bad;
    fclose(f);
    return 0;
bad:
    fclose(f);
    orc_free_input(in);
    return 2;
}

void orc_free_input(orc_input *in) {
    free(in->start); free(in->X); free(in->y);
    memset(in, 0, sizeof(*in));
}

/* ----------------------------------------------------------- whole pipeline */

int orc_linreg(const orc_input *in, int p1, int p2, int w1, int w2, int alg, int iters,
               double lambda, int64_t *beta) {
    size_t n = in->n, d = in->d, T = d * (d + 1) / 2;
    if (p2 < 0) p2 = p1;
    int64_t *Xq = malloc(n * d * 8), *yq = malloc(n * 8);
    uint64_t *A = malloc(T * 8), *b = malloc(d * 8);
    int64_t *a2 = malloc(T * 8), *b2 = malloc(d * 8);
    orc_quantize(in->X, n * d, p1, n, w2, Xq);
    orc_quantize(in->y, n, p1, n, w2, yq);
    orc_aggregate(Xq, yq, n, d, p1, w1, A, b);
    if (w1 == 64 && w2 == 32) {
        /* single-share view of phase1.c:609-638 (exact only when one DP holds everything) */
        orc_convert_shares(A, T, p1, p2, w1, w2, A);
        orc_convert_shares(b, d, p1, p2, w1, w2, b);
    }
    orc_sum_shares(A, 1, T, w2, a2);
    orc_sum_shares(b, 1, d, w2, b2);
    orc_circuit_input(a2, b2, d, lambda, p2, w2);
    if (alg == 0) orc_cholesky(a2, b2, d, p2, w2, beta);
    else if (alg == 1) orc_ldlt(a2, b2, d, p2, w2, beta);
    else orc_cgd(a2, b2, d, p2, w2, iters, beta, NULL);
    free(Xq); free(yq); free(A); free(b); free(a2); free(b2);
    return 0;
}
