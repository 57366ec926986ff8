/*
 * linreg_gc.h -- C ABI of the MI355X garbled-circuit engine (liblinreg_gc.so).
 *
 * Drop-in boundary for the garbled-circuit hot path of schoppmp/linreg-mpc.
 * Plain C types only; the library owns all device memory; every function
 * returns 0 on success and a negative LGC_E* code otherwise (never aborts the
 * process; the reference's convention is check()/goto error -> exit 1,
 * src/check_error.h:5-9).  lgc_last_error() gives the message.
 *
 * What each entry point replaces in the reference:
 *
 *   lgc_solver_*            execYaoProtocol(pd, cholesky|ldlt|cgd, &ls)
 *                           (src/cmd/linreg.c:177, src/cmd/test/test_linear_system.c:125)
 *                           with linear_system_t (src/linear.h:16-26) as lgc_system
 *   lgc_solver_set_shares   the circuit-input side of dcrRecvIntArray / feedOblivLLong
 *                           (src/input.c:81-113, src/linear.oc:31-49, :116-127)
 *   lgc_agg_*               inner_product_local and the diagonal special case
 *                           (src/phase1.c:14-20, 562-571)
 *   lgc_ti_*                run_trusted_initializer / inner_product_ti arithmetic
 *                           (src/phase1.c:148-236, 241-339)
 *   lgc_ot_*                honest[Correlated]OTExt{Send,Recv}1Of2 (IKNP) as used at
 *                           src/phase1.c:58-65,84-89 and src/input.c:44,108
 *
 * The library fails loudly (LGC_ENODEVICE) when no HIP device is present:
 * there is no CPU fallback.
 */
#ifndef LINREG_GC_H
#define LINREG_GC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGC_OK 0
#define LGC_EINVAL (-1)
#define LGC_ENODEVICE (-2)
#define LGC_EHIP (-3)
#define LGC_ENOMEM (-4)
#define LGC_ESTATE (-5)

#define LGC_ALG_CHOLESKY 0
#define LGC_ALG_LDLT 1
#define LGC_ALG_CGD 2

const char *lgc_last_error(void);
int lgc_device_count(void);
/* version / build string */
const char *lgc_version(void);

/* ------------------------------------------------------------------ phase 2 */

/* counterpart of linear_system_t (src/linear.h:16-26) */
typedef struct {
    size_t d;              /* a.d[0] == a.d[1] == b.len */
    int width;             /* FIXED_BIT_SIZE_P2: 32 or 64 (src/fixed.h:17-30); runtime here */
    int precision;         /* ls.precision */
    int algorithm;         /* LGC_ALG_* (argv[4] of bin/linreg) */
    int num_iterations;    /* ls.num_iterations (cgd) */
    double lambda;         /* ls.lambda */
    size_t nshares;        /* number of additive input shares per entry (DPs, or 2) */
    int normalize;         /* 1: data-provider path, diag += lambda and /d (linear.oc:52-65);
                              0: two-party path (linear.oc:96-135) */
    int reveal_inputs;     /* debug reveal of A, b (linear.oc:68-84) */
    int trace;             /* per-iteration reveals of x, gamma, eta, q, ng (cgd.oc:167-189) */
} lgc_system;

typedef struct {
    uint64_t and_gates;        /* this build's non-free gate count (yaoGateCount analogue) */
    uint64_t gate_steps;       /* wave-level gate steps (64 lanes each) */
    uint64_t table_bytes;      /* garbled-table bytes streamed garbler -> evaluator */
    uint64_t launches;
    double seconds_total;      /* input labels + garble + evaluate + decode, device time */
    double seconds_garble;     /* sum over garble kernels (0 unless profiling is on) */
    double seconds_eval;       /* sum over evaluate kernels */
    double seconds_mac_garble; /* the dominant kernel: garbling of the MAC launches */
    double seconds_mac_eval;
    uint64_t mac_gates;        /* AND gates inside MAC launches */
    uint64_t mac_launches;
} lgc_stats;

typedef struct lgc_solver lgc_solver;

/* Builds the circuit program and allocates device memory on `device`.
 * seed: 16 bytes of garbler randomness (global offset R and input labels). */
int lgc_solver_create(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16]);
void lgc_solver_destroy(lgc_solver *s);

/* shares: nshares x (T + d) words, share-major: [A packed lower triangle (T = d(d+1)/2,
 * idx(i,j) = i(i+1)/2 + j, src/linear.c:11-16)] then [b (d)], low `width` bits used. */
int lgc_solver_set_shares(lgc_solver *s, const uint64_t *shares);

/* Garble + evaluate the whole circuit (both roles co-located on this GPU).
 * profile != 0 brackets every kernel with HIP events (slower; fills the per-kernel times). */
int lgc_solver_run(lgc_solver *s, int profile);

/* Results (sign-extended to int64 when width == 32).  beta: d.
 * trace: num_iterations x (d + 4) (x, gamma, eta, q, ng), inputs: T + d. */
int lgc_solver_get_beta(lgc_solver *s, int64_t *beta);
int lgc_solver_get_trace(lgc_solver *s, int64_t *trace);
int lgc_solver_get_inputs(lgc_solver *s, int64_t *ab);
int lgc_solver_get_stats(lgc_solver *s, lgc_stats *st);
/* per-launch kernel times of the last profiled run (seconds; n = number of launches) */
int lgc_solver_get_profile(lgc_solver *s, double *garble_s, double *eval_s, size_t n);

/* Introspection (host only, no GPU needed): the lowered program.  Used by the
 * CPU test-suite to run the very same records on the CPU checker. */
typedef struct {
    uint32_t op, cnt, dst, a, b, c;
    int32_t sa, sb;
    uint64_t step0;
} lgc_record;
typedef struct {
    uint32_t first_rec, nrec;
    uint64_t step0, steps, gates;
    int mac_only;
} lgc_launch;
typedef struct {
    size_t n_records, n_launches;
    uint32_t n_words, n_reveal, in_base, rv_beta, rv_trace, rv_inputs;
    uint64_t total_steps, total_gates, max_launch_steps;
} lgc_program_info;
typedef struct lgc_program lgc_program;
int lgc_program_build(lgc_program **out, const lgc_system *sys);
void lgc_program_destroy(lgc_program *p);
int lgc_program_info_get(const lgc_program *p, lgc_program_info *info);
const lgc_record *lgc_program_records(const lgc_program *p);
const lgc_launch *lgc_program_launches(const lgc_program *p);

/* One-shot convenience: create + set + run + get + destroy. */
int lgc_solve(int device, const lgc_system *sys, const uint8_t seed[16], const uint64_t *shares,
              int64_t *beta, int64_t *trace, lgc_stats *stats);

/* ------------------------------------------------------- micro-benchmarks */
/* Stand-alone LDS T-table AES throughput (the "AES roofline" of the north
 * star): blocks_per_lane AES-128 encryptions in every lane of `waves` waves.
 * Returns blocks/second in *rate; *check gets an XOR checksum. */
int lgc_aes_bench(int device, int waves, int blocks_per_lane, double *rate, uint32_t *check);
/* AES-128 of `n` 16-byte blocks with the fixed key on the device (known-answer tests). */
int lgc_aes_encrypt(int device, const uint8_t *in, uint8_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
