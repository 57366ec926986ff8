/*
 * linreg_oracle.h -- CPU *semantic* oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-integer restatement of the reference's fixed-point pipeline
 * (schoppmp/linreg-mpc): quantisation, phase-1 aggregation, circuit input
 * assembly, the ofixed_* primitives and the cgd / cholesky / ldlt solvers.
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product path (linreg-mpc_amd/) never does.
 *
 * Parity pin: tests/test_oracle_readme_kat.py checks this oracle against the
 * only input->output pair the reference ships (README.md:51-88,
 * examples/readme_example.in): all five printed coefficients to 15 decimals.
 * Cholesky / LDL^T outputs, 64-bit sqrt, and division by zero have no known
 * answer in the reference: "parity unpinned" for those.
 *
 * Conventions: W in {32,64} is the two's-complement width; values travel as
 * int64_t (sign-extended when W == 32).  p = precision (fractional bits).
 */
#ifndef LINREG_ORACLE_H
#define LINREG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar primitives (src/fixed.c, src/fixed.oc) ---- */
int64_t orc_wrap(int64_t v, int w);                         /* reduce to W-bit two's complement */
int64_t orc_double_to_fixed(double d, int p, int w);        /* fixed.c:3-5,11-13 */
double  orc_fixed_to_double(int64_t f, int p);              /* fixed.c:7-9,15-17 */
int64_t orc_add(int64_t a, int64_t b, int w);               /* fixed.oc:99-109 */
int64_t orc_sub(int64_t a, int64_t b, int w);               /* fixed.oc:111-121 */
int64_t orc_abs(int64_t a, int w);                          /* fixed.oc:90-97 */
int     orc_cmp(int64_t a, int64_t b, int w);               /* fixed.oc:78-88 */
int64_t orc_mul(int64_t a, int64_t b, int p, int w);        /* fixed.oc:149-162 */
int64_t orc_div(int64_t a, int64_t b, int p, int w);        /* fixed.oc:164-188 */
int64_t orc_sqrt(int64_t a, int p, int w);                  /* fixed.oc:217-248 */
int64_t orc_inner_product(const int64_t *a, const int64_t *b, size_t d, int p, int w); /* fixed.oc:124-147 */
size_t  orc_idx(size_t i, size_t j);                        /* linear.c:11-16 */

/* ---- quantisation (src/phase1.c:473-476, src/linear.c:46-51,82-87) ---- */
/* out[k] = (fixed_t)((vals[k] / sqrt(2^p1 * n)) * 2^p1), cast through the
 * phase-2 type (w2) as the reference does, stored sign-extended. */
void orc_quantize(const double *vals, size_t count, int p1, size_t n, int w2, int64_t *out);

/* ---- phase-1 aggregate: the value the shares must sum to (phase1.c:14-20,534-588) ---- */
/* Xq: n x d row-major, yq: n.  A: packed lower triangle T = d(d+1)/2, b: d.
 * Off-diagonal and b wrap mod 2^w1; the diagonal follows the double path of
 * phase1.c:562-567 (already divided by d). */
void orc_aggregate(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                   int p1, int w1, uint64_t *A, uint64_t *b);

/* ---- phase-1 share-level simulation, TI mode (phase1.c:148-236,241-339,534-588) ---- */
/* start[k] = first column owned by DP k (k < P); the last DP owns y.
 * rnd: stream of uniformly random w1-bit words consumed exactly as the TI
 * does (x[n], y[n], r per cross-party pair in (i, j) order).  rnd_words in:
 * capacity; out: words consumed.  shares_A: P x T, shares_b: P x d. */
/* words [first, first + count) of the TI's random stream -> out; non-zero = stream exhausted */
typedef int (*orc_rnd_fn)(void *ctx, size_t first, size_t count, uint64_t *out);
int orc_phase1_ti_shares_cb(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                            int p1, int w1, size_t P, const size_t *start,
                            orc_rnd_fn next, void *ctx, size_t *rnd_words,
                            uint64_t *shares_A, uint64_t *shares_b);
int orc_phase1_ti_shares(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                         int p1, int w1, size_t P, const size_t *start,
                         const uint64_t *rnd, size_t *rnd_words,
                         uint64_t *shares_A, uint64_t *shares_b);

/* ---- phase-1 share-level simulation, OT mode (phase1.c:38-96,353-450) ---- */
/* Gilboa product sharing.  rnd supplies the sender's s values (one w1-bit
 * word per OT, consumed in the order the threads' (i, j) loops run for the
 * pair, lower DP index first). */
int orc_phase1_ot_shares(const int64_t *Xq, const int64_t *yq, size_t n, size_t d,
                         int p1, int w1, size_t P, const size_t *start,
                         const uint64_t *rnd, size_t *rnd_words,
                         uint64_t *shares_A, uint64_t *shares_b);

/* 64 -> 32 conversion of one share vector (phase1.c:609-638). */
void orc_convert_shares(const uint64_t *in, size_t count, int p1, int p2, int w1, int w2, uint64_t *out);

/* ---- circuit input assembly (linear.oc:18-65) ---- */
/* shares: P x count (count = T for A, d for b), summed mod 2^w2. */
void orc_sum_shares(const uint64_t *shares, size_t P, size_t count, int w2, int64_t *out);
/* diag += (fixed_t)(lambda*2^p); off-diag and b: C division by d. In place. */
void orc_circuit_input(int64_t *a, int64_t *b, size_t d, double lambda, int p, int w);

/* ---- solvers (cgd.oc:96-212, cholesky.oc:51-93, ldlt.oc:50-90) ---- */
/* a: packed lower triangle (T), b: d.  beta: d out.
 * trace (may be NULL): iters x (d + 4) words: x[0..d), gamma, eta, q, ng
 * exactly as revealed per iteration at cgd.oc:167-189. */
void orc_cgd(const int64_t *a, const int64_t *b, size_t d, int p, int w, int iters,
             int64_t *beta, int64_t *trace);
void orc_cholesky(const int64_t *a, const int64_t *b, size_t d, int p, int w, int64_t *beta);
void orc_ldlt(const int64_t *a, const int64_t *b, size_t d, int p, int w, int64_t *beta);

/* ---- input file (README.md:51-77, config.c:24-44, linear.c:33-36,73) ---- */
typedef struct {
    size_t n, d, P;
    size_t *start;      /* P entries */
    double *X;          /* n*d row-major */
    double *y;          /* n */
} orc_input;
int  orc_read_input(const char *path, orc_input *in);
void orc_free_input(orc_input *in);

/* ---- whole pipeline: what bin/linreg party 2 prints as "Result:" (linreg.c:181-187) ---- */
/* alg: 0 cholesky, 1 ldlt, 2 cgd.  p2 < 0 means "same as p1".
 * Uses the aggregate totals (share randomness only matters when w1 != w2;
 * for that case use the share-level functions). */
int orc_linreg(const orc_input *in, int p1, int p2, int w1, int w2, int alg, int iters,
               double lambda, int64_t *beta);

#ifdef __cplusplus
}
#endif
#endif
