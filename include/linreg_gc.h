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
 *   lgc_p1_local            inner_product_local and the diagonal special case
 *                           (src/phase1.c:14-20, 562-571)
 *   lgc_p1_* / lgc_ti_generate
 *                           inner_product_ti arithmetic / run_trusted_initializer
 *                           (src/phase1.c:148-236, 241-339)
 *   lgc_ot_*                honest[Correlated]OTExt{Send,Recv}1Of2 (IKNP) as used at
 *                           src/phase1.c:58-65,84-89 and src/input.c:44,108
 *
 * INTEGRATION.md names the reference call behind every one of the 70 entry points of this header.  Two more headers
 * complete the library's exports and are NOT part of the drop-in surface:
 *   linreg_gc_sweep.h   the per-lambda sweep (BASELINE config 5): one program for many regularisation
 *                       values, its shared prefix, the blocks of a sweep sharded over several GPUs
 *   linreg_gc_debug.h   tracing, introspection of the lowered program, start-up and tuning hints, test hooks,
 *                       micro-benchmarks
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
/* Not a solver: "check if inputs have equal dimensions" of the two-party input path (src/linear.oc:109-114,
 * revealOblivBool(feedOblivInt(d, 1) == feedOblivInt(d, 2))) as a program of its own -- d = 1, nshares = 2, normalize = 0,
 * width 32; the first input word of share 0 / share 1 is party 1's / party 2's dimension, beta[0] is 1 when they are equal
 * (31 AND gates).  Run before the solve, whose program both parties can only build once they agree on d. */
#define LGC_ALG_DIMCHECK 3

const char *lgc_last_error(void);
int lgc_device_count(void);
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

/* Results (sign-extended to int64 when width == 32).  beta: d (sweep: count x d).
 * trace: num_iterations x (d + 4) (x, gamma, eta, q, ng), inputs: T + d. */
int lgc_solver_get_beta(lgc_solver *s, int64_t *beta);
int lgc_solver_get_trace(lgc_solver *s, int64_t *trace);
int lgc_solver_get_inputs(lgc_solver *s, int64_t *ab);
int lgc_solver_get_stats(lgc_solver *s, lgc_stats *st);
/* cgd only, n = num_iterations: AND gates emitted and device seconds elapsed (since the start of
 * lgc_solver_run) when iteration t's reveals were evaluated -- the values cgd.oc:190-194 prints as
 * "Iteration t gate count" / "Iteration t time".  Either output pointer may be NULL. */
int lgc_solver_get_iterations(lgc_solver *s, uint64_t *and_gates, double *seconds, size_t n);

/* ----------------------------------------- phase 2 with the two roles apart */
/* CSP (party 1, garbler) and Evaluator (party 2) as separate objects, possibly in different
 * processes (src/cmd/linreg.c:145-199).  The host carries the bytes: input labels (through the OT
 * below, src/input.c), one table buffer per launch, the garbler's decode bits at the end.
 * Both sides must pass the same `sys` and `max_launch_table_bytes` (0 = 256 MiB): the lowered
 * program, and with it every table offset and gate id, is a function of those only. */
#define LGC_ROLE_GARBLER 1
#define LGC_ROLE_EVALUATOR 2
typedef struct lgc_party lgc_party;
int lgc_party_create(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                     size_t max_launch_table_bytes);
void lgc_party_destroy(lgc_party *p);
size_t lgc_party_num_launches(const lgc_party *p);
size_t lgc_party_table_bytes(const lgc_party *p, size_t launch);
size_t lgc_party_input_bits(const lgc_party *p);      /* (T + d) * width, per share */
size_t lgc_party_num_reveal(const lgc_party *p);
uint64_t lgc_party_and_gates(const lgc_party *p);
/* 32 bytes over everything the two roles of a solve must have in common (records incl. lambda and gate-step numbers, launch
 * boundaries, width, precision).  The host binaries exchange and compare it before the first table moves, so that
 * an option given to one party only (--lambdas, --prec_phase2 ...) is an error message instead of a wrong
 * result.  A check against misconfiguration, not an authentication of the peer. */
int lgc_party_program_fingerprint(const lgc_party *p, uint8_t out[32]);
/* cgd only, n = num_iterations: the launch that completes iteration t and the AND gates emitted up
 * to and including it (cgd.oc:190-194 prints both per iteration).  Either pointer may be NULL. */
int lgc_party_iteration_marks(const lgc_party *p, uint32_t *launch, uint64_t *and_gates, size_t n);
/* garbler: per input bit of share k (word-major, LSB first: sel[i*intsize+j], src/input.c:41) the
 * label pair (m0, m1), 16 bytes each -- the sender messages of dcrRecvBitArray's OT (input.c:94-108) */
int lgc_party_input_pairs(lgc_party *p, size_t share, uint8_t *m0, uint8_t *m1);
/* garbler: labels of values it knows itself (feedOblivLLong for its own party, linear.oc:116-127) */
int lgc_party_encode_inputs(lgc_party *p, size_t share, const uint64_t *values, uint8_t *labels_out);

/* Device-resident table hand-off for a garbler and an evaluator PROCESS on the same node (same MI355X, or two GPUs of one
 * xGMI hive; bin/linreg --table_ring): instead of the osend/orecv byte stream that carries the garbled tables in the
 * reference (Obliv-C Yao runtime under execYaoProtocol, linreg.c:177) the garbler allocates a BYTE ring in its HBM --
 * `ring_bytes` (0 = the library's choice: the largest launch plus half as much again, the extra between 64 MiB and 4 GiB) in
 * which launch k owns a contiguous range, laid out identically in both processes (same program, same size, same offsets) --,
 * exports it as a 64-byte hipIpc handle (sent over the existing socket), and the evaluator maps it.  The host keeps the
 * ordering: before lgc_party_garble_ring(k) the garbler's host waits until the evaluator has finished launch
 * lgc_party_ring_wait_for(p, k) (-1: nothing to wait for) -- the newest earlier launch whose range launch k overwrites; the
 * evaluator may call lgc_party_evaluate_ring(k) only after lgc_party_garble_ring(k) has returned (or _wait(k), below).  Both
 * calls block until their kernel has completed.
 * What the ring ever holds is what the socket would carry: garbled tables (and zeros).  The garbler's intermediate state of
 * critical-path launches -- zero-labels, from which R follows -- stays in a buffer private to its process (src/input.c:94-108:
 * label pairs never leave the CSP); a failed call leaves the range with its previous contents. */
int lgc_party_ring_create_bytes(lgc_party *p, size_t ring_bytes, uint8_t handle_out[64], size_t *ring_bytes_out);
int lgc_party_ring_open_bytes(lgc_party *p, const uint8_t handle[64], size_t ring_bytes);
int64_t lgc_party_ring_wait_for(const lgc_party *p, size_t launch);
int lgc_party_garble_ring(lgc_party *p, size_t launch);
int lgc_party_evaluate_ring(lgc_party *p, size_t launch);
/* lgc_party_garble_ring in two halves (round 5), for a garbler process that does not stop after every launch -- where the
 * reference's Yao runtime keeps garbling while the bytes of earlier gates are still on their way (osend under
 * execYaoProtocol, src/cmd/linreg.c:177): _begin(k) enqueues launch k behind launch k - 1 and returns at once (the table pass
 * of a critical-path launch runs on a stream of its own, beside the next launches' record kernels; the zero-label stash
 * alternates between two private buffers); _wait(k) returns once the tables of launch k are complete in the ring, i.e. when
 * the evaluator may be told.  The ring discipline is the caller's as before (lgc_party_ring_wait_for(k) acknowledged before
 * _begin(k)); launches are waited for in order, at most 63 of them begun and not yet waited for; _wait may be called from a
 * second thread while the first goes on enqueueing (host/tables.c does).  What the ring ever holds is unchanged. */
int lgc_party_garble_ring_begin(lgc_party *p, size_t launch);
int lgc_party_garble_ring_wait(lgc_party *p, size_t launch);
/* evaluator: the labels a data provider forwarded (dcsSendIntArray -> orecv, input.c:46, 84-92) */
int lgc_party_set_input_labels(lgc_party *p, size_t share, const uint8_t *labels);
/* The same two hand-overs with the labels left in HBM (bin/linreg --input_ring: all parties on one node).  The garbler exports a
 * share's label pairs into two device buffers of its own (input_bits x 16 bytes each; never shared, cleared with
 * lgc_dev_free_secret) which its OT sender session reads in place (lgc_ot_sender_set_device_io); the evaluator imports the
 * chosen labels from device memory, e.g. a buffer of the data provider mapped with lgc_dev_open. */
int lgc_party_input_pairs_dev(lgc_party *p, size_t share, void *d0, void *d1);
int lgc_party_set_input_labels_dev(lgc_party *p, size_t share, const void *dev_labels);
int lgc_party_garble(lgc_party *p, size_t launch, uint8_t *tables_out);
int lgc_party_evaluate(lgc_party *p, size_t launch, const uint8_t *tables_in);
int lgc_party_decode_bits(lgc_party *p, uint64_t *dec_out);       /* garbler -> evaluator */
int lgc_party_finish(lgc_party *p, const uint64_t *garbler_dec, int64_t *beta, int64_t *trace, int64_t *inputs);

/* Device buffers for C host code: zero-filled device memory, optionally with a 64-byte hipIpc handle that a
 * peer process on the same node opens (same GPU, or an xGMI peer) -- the hand-off of the OT extension's
 * u / y between two data providers without the socket (bin/linreg --ot_ring), next to the table ring. */
int lgc_dev_alloc(int device, size_t bytes, void **ptr, uint8_t handle_out[64] /* or NULL */);
void lgc_dev_free(void *ptr);
void lgc_dev_free_secret(void *ptr, size_t bytes);     /* zero-filled first */
int lgc_dev_open(int device, const uint8_t handle[64], void **ptr);
void lgc_dev_close(void *ptr);
int lgc_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int lgc_dev_download(void *dst_host, const void *src_dev, size_t bytes);

/* ------------------------------------------------------------------ phase 1 */
/* Quantised data of one data provider, resident on the device (src/phase1.c:473-476 result). */
typedef struct lgc_p1 lgc_p1;
int lgc_p1_create(lgc_p1 **out, int device, size_t n, size_t d, int width, int precision);
void lgc_p1_destroy(lgc_p1 *h);
/* Xq: n x d row-major fixed point (sign-extended when width == 32); yq: n or NULL. */
int lgc_p1_set_data(lgc_p1 *h, const int64_t *Xq, const int64_t *yq);
/* Shares a data provider computes alone for its own columns [c0, c1): out_A is the packed lower
 * triangle of that block (local idx(i,j), (c1-c0)(c1-c0+1)/2 words): inner_product_local off the
 * diagonal (src/phase1.c:14-20, 569-571), the double-precision special case on it (562-567).
 * with_y: also out_b[i] = <column c0+i, y> (the last data provider, phase1.c:376-381). */
int lgc_p1_local(lgc_p1 *h, size_t c0, size_t c1, int with_y, uint64_t *out_A, uint64_t *out_b);
/* inner_product_ti arithmetic, batched over pairs (src/phase1.c:148-236); column index d means y.
 * lgc_p1_mask: out[q][k] = column cols[q] + sign * V[q][k]      (b + x: sign +1; a - y: sign -1)
 * lgc_p1_dot : out[q] = <A[q], B[q]> (colsB NULL) or <A[q], column colsB[q]>, minus sub[q] */
int lgc_p1_mask(lgc_p1 *h, const uint32_t *cols, size_t npairs, const uint64_t *V, int sign, uint64_t *out);
int lgc_p1_dot(lgc_p1 *h, const uint64_t *A, const uint64_t *B, const uint32_t *colsB, size_t npairs,
               const uint64_t *sub, uint64_t *out);
/* party a of ONE inner_product_ti (src/phase1.c:171-197) in a single pass over the column:
 * out_mask[n] = a - y (the message for party b) and *share = <in, y> - sub, where `in` is party b's
 * message b + x and (y, sub) came from the trusted initializer.  Same results as lgc_p1_mask(sign
 * -1) followed by lgc_p1_dot, with half the device operations. */
int lgc_p1_ti_a(lgc_p1 *h, uint32_t col, const uint64_t *y, const uint64_t *in, uint64_t sub,
                uint64_t *out_mask, uint64_t *share);
/* the same for a run of pairs with one peer in ONE device call (y, in, out_mask: npairs x n words) */
int lgc_p1_ti_a_batch(lgc_p1 *h, const uint32_t *cols, size_t npairs, const uint64_t *y, const uint64_t *in,
                      const uint64_t *sub, uint64_t *out_mask, uint64_t *shares);
/* Device I/O for the three calls above: V / out, A / B, y / in / out_mask are device memory used in place
 * (cols, sub and the returned shares stay host arrays) -- the TI-mode exchange between parties on one
 * node through device rings (bin/linreg --ti_ring). */
int lgc_p1_set_device_io(lgc_p1 *h, int on);
/* Trusted initializer (src/phase1.c:241-287): pairs [first_pair, first_pair + npairs) of the
 * cross-party (i, j) enumeration; x, y: npairs x n words, r, xy_minus_r: npairs words, drawn in the
 * order x, y, r from one AES-128-CTR stream keyed by seed (newBCipherRandomGen / randomizeBuffer). */
int lgc_ti_generate(int device, const uint8_t seed[16], uint64_t first_pair, size_t npairs, size_t n, int width,
                    uint64_t *x, uint64_t *y, uint64_t *r, uint64_t *xy_minus_r);

/* as lgc_ti_generate, with x[q] / y[q] written to the device addresses x_dst[q] / y_dst[q] */
int lgc_ti_generate_scatter(int device, const uint8_t seed[16], uint64_t first_pair, size_t npairs, size_t n, int width,
                            void *const *x_dst, void *const *y_dst, uint64_t *r, uint64_t *xy_minus_r);

/* ------------------------------------------------------------- OT extension */
/* IKNP semi-honest OT extension, kappa = 128.  The base OTs (Naor-Pinkas in Obliv-C) run on the
 * host; their outputs come in as seeds: the extension sender holds delta and k_j^{delta_j}, the
 * extension receiver holds (k_j^0, k_j^1).  Replaces honestOTExtSenderNew / RecverNew and the
 * Send1Of2 / Recv1Of2 / correlated variants (src/phase1.c:58-65,84-89,394-397; src/input.c:28-44,67,108).
 * A transfer is three calls: receiver *_recv_start -> u (16 bytes per OT, lgc_ot_u_bytes(m));
 * sender *_send(u) -> payload; receiver *_recv_finish(payload). */
typedef struct lgc_ot_sender lgc_ot_sender;
typedef struct lgc_ot_receiver lgc_ot_receiver;
int lgc_ot_sender_create(lgc_ot_sender **out, int device, const uint8_t delta[16], const uint8_t seeds[128][16]);
void lgc_ot_sender_destroy(lgc_ot_sender *s);
int lgc_ot_receiver_create(lgc_ot_receiver **out, int device, const uint8_t seeds0[128][16], const uint8_t seeds1[128][16]);
void lgc_ot_receiver_destroy(lgc_ot_receiver *r);
size_t lgc_ot_u_bytes(uint64_t m);
/* Device I/O: with on != 0 every data pointer of the transfer calls of this session (a, b, choice, u, y, e,
 * messages, labels, shares) is DEVICE memory on the session's GPU and is used in place -- no host round
 * trip between lgc_p1_* outputs, the OT and the consumer.  Default: host pointers (copied on the
 * session's stream; page-locked buffers from lgc_host_alloc move at the full PCIe rate). */
int lgc_ot_sender_set_device_io(lgc_ot_sender *s, int on);
int lgc_ot_receiver_set_device_io(lgc_ot_receiver *r, int on);
/* Gilboa inner products (inner_product_ot_recver / _sender, src/phase1.c:53-96), batched over
 * npairs: OT index (q*n + k)*width + bit; receiver choice = bit of a[q][k] (LSB first), sender
 * correlation 2^bit * b[q][k] + s.  y: npairs*n*width words.  shares: npairs words each side;
 * share_sender + share_receiver = <a[q], b[q]> mod 2^width.
 * Up to four receives may be in flight on one receiver: every *_recv_finish completes the OLDEST
 * started batch (the sender answers in order), and one thread may call *_recv_start while another
 * calls *_recv_finish -- the network round trip of batch k then overlaps the extension of batch k+1. */
int lgc_ot_gilboa_recv_start(lgc_ot_receiver *r, const uint64_t *a, size_t npairs, size_t n, int width, uint8_t *u_out);
int lgc_ot_gilboa_send(lgc_ot_sender *s, const uint64_t *b, size_t npairs, size_t n, int width, const uint8_t *u_in,
                       uint64_t *y_out, uint64_t *shares);
int lgc_ot_gilboa_recv_finish(lgc_ot_receiver *r, const uint64_t *y_in, uint64_t *shares);
/* 1-of-2 OT of 16-byte messages (wire labels; dcsSendIntArray / dcrRecvBitArray, src/input.c:37-113).
 * choice: one byte per OT; e: 32 bytes per OT; out: 16 bytes per OT. */
int lgc_ot_labels_recv_start(lgc_ot_receiver *r, const uint8_t *choice, size_t m, uint8_t *u_out);
int lgc_ot_labels_send(lgc_ot_sender *s, const uint8_t *msg0, const uint8_t *msg1, size_t m, const uint8_t *u_in, uint8_t *e_out);
int lgc_ot_labels_recv_finish(lgc_ot_receiver *r, const uint8_t *e_in, uint8_t *out);

/* Page-locked host memory for buffers that are handed to the entry points above (tables, OT
 * messages, phase-1 vectors): the copies to and from the GPU then run as DMA at PCIe speed instead
 * of through a pageable staging copy.  Optional -- every entry point accepts ordinary memory.
 * Returns NULL (and sets lgc_last_error) on failure. */
void *lgc_host_alloc(size_t bytes);
void lgc_host_free(void *p);
#ifdef __cplusplus
}
#endif
#endif
