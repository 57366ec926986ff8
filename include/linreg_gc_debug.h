/*
 * linreg_gc_debug.h -- exports of liblinreg_gc.so that are NOT part of the drop-in surface (linreg_gc.h): tracing,
 * introspection of the lowered program, start-up and tuning hints, test hooks, micro-benchmarks.  The host binaries use
 * the tracing, start-up and slot-ring calls; tests/ and bench.py use the rest.
 */
#ifndef LINREG_GC_DEBUG_H
#define LINREG_GC_DEBUG_H
#include "linreg_gc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- version, tracing */
/* version / build string */
const char *lgc_version(void);
/* LINREG_TRACE=1 in the environment: wall-clock marks on stderr, one line each, "LGCT <tag> <CLOCK_MONOTONIC seconds> <what>"
 * -- the library marks its own start-up steps (HIP runtime, device context, program lowered, buffers, table ring), a host
 * adds its protocol steps with the same call, and since the clock is system-wide the marks of all parties of a run line up
 * (bench.py: `phase12[].timeline`).  Replaces nothing in the reference; it is how the start-up of src/cmd/linreg.c:100-199
 * is broken down here.  Without the variable both calls do nothing. */
void lgc_trace_set_tag(const char *tag);
void lgc_trace_mark(const char *what);

/* ---- start-up hints (optional; bin/linreg calls them from a thread while it parses and connects) */
/* Brings the HIP runtime and the context of `device` up and issues a first dispatch (60-250 ms when several parties start
 * together, 20-50 ms for the first dispatch); all of it is process-wide, so a host may call this from a thread while it
 * parses its input and connects (bin/linreg does). */
int lgc_device_warm(int device);
/* Loads code objects and creates streams ahead of their first use (a code object is otherwise loaded inside the first launch of
 * one of its kernels, 5-10 ms; a stream costs ~10 ms): what & 1 the phase-1 kernels, what & 2 the OT kernels, what & 4 two
 * streams for the pool the OT sessions draw from.  lgc_party_create* preloads the record kernels of its program by itself. */
int lgc_preload(int device, int what);

/* per-launch kernel times of the last profiled run (seconds; n = number of launches) */
int lgc_solver_get_profile(lgc_solver *s, double *garble_s, double *eval_s, size_t n);

/* ---- the lowered program (host only, no GPU needed) */
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
    uint32_t replicas, word_stride, reveal_stride;   /* sweep programs: circuit t uses words x + t * word_stride
                                                        (x >= shared_end) and decode slots r + t * reveal_stride */
    uint32_t shared_end, prefix_launches;            /* words [0, shared_end) and launches [0, prefix_launches) are */
    uint64_t prefix_steps;                           /* the lambda-independent prefix (inputs, share sums, normalizer) */
    uint64_t total_xors;                             /* XOR gates a flat gate list of this circuit would hold (word XORs x width): */
                                                     /* reporting only (bytes = 192 N_AND + 128 N_XOR)            */
} lgc_program_info;
typedef struct lgc_program lgc_program;
int lgc_program_build(lgc_program **out, const lgc_system *sys);
int lgc_program_build_sweep(lgc_program **out, const lgc_system *sys, size_t count, const double *lambdas);
/* the block [first, first + count) of a larger sweep (gate ids offset by `first` circuits) */
int lgc_program_build_sweep_at(lgc_program **out, const lgc_system *sys, size_t count, const double *lambdas, size_t first);
void lgc_program_destroy(lgc_program *p);
int lgc_program_info_get(const lgc_program *p, lgc_program_info *info);
const lgc_record *lgc_program_records(const lgc_program *p);
const lgc_launch *lgc_program_launches(const lgc_program *p);
/* The co-located solver's garbled-table ring for this program: launch i owns bytes
 * [offsets[i], offsets[i] + 2048 * steps_i) (rounded up to 4 KiB) of a ring of *ring_bytes_out; the
 * garbler may overwrite that range once launch wait_for[i] (-1: nobody) has been evaluated.
 * ring_bytes = 0 asks for the solver's own choice (twice the largest launch).  Arrays: n_launches. */
int lgc_program_ring_plan(const lgc_program *p, size_t ring_bytes, size_t *ring_bytes_out, size_t *offsets, int64_t *wait_for);

/* The gate count the REFERENCE's own circuit has for this solve (two-party input path; exact fits to every
 * result file under experiments/results/phase2_32 and phase2_64) -- this build's circuits are smaller, so results files and
 * rates carry both counts (bin/test_linear_system prints it, python/results.py writes it as an extra column).
 * cgd: the cumulative count after `iterations` iterations.  LGC_EINVAL for ldlt (nothing published).  Host only. */
int lgc_reference_gate_count(int algorithm, int width, size_t d, int iterations, uint64_t *gates);

/* One-shot convenience: create + set + run + get + destroy. */
int lgc_solve(int device, const lgc_system *sys, const uint8_t seed[16], const uint64_t *shares,
              int64_t *beta, int64_t *trace, lgc_stats *stats);

/* ---- tuning and legacy forms */
/* The table ring in its first form: `nslots` slots of the largest launch's size, launch k in slot k % nslots (the garbler may
 * call lgc_party_garble_ring(k) only after the evaluator has finished launch k - nslots).  Superseded by the byte ring of
 * linreg_gc.h (a slot ring of 4 x the largest launch was 10.5 GB at d = 100 and 33 GB for config 4); bin/linreg
 * --table_ring=<nslots> and the ring tests still use it. */
int lgc_party_ring_create(lgc_party *p, int nslots, uint8_t handle_out[64], size_t *slot_bytes);
int lgc_party_ring_open(lgc_party *p, const uint8_t handle[64], int nslots, size_t slot_bytes);
/* streams = 1: table passes stay on the record kernels' stream (no second queue to create, no second stash): all that is
 * left of the asynchronous path is that the garbler's stream does not wait for the host between launches -- the form for
 * short programs, where creating a queue costs more than overlapping the passes gains.  2 (default): as above.  Before the
 * first _begin. */
int lgc_party_garble_ring_streams(lgc_party *p, int streams);
/* A destroyed lgc_solver leaves its table ring (its one large device allocation) parked for the next
 * solver on that device: allocating tens of GB right after freeing as much costs more than a solve.
 * This call frees what is parked (all devices). */
void lgc_release_cached_memory(void);

/* Kernel choice for latency-bound launches (at most one record per CU), per role: non-zero (the
 * default) runs them column-split on 16 waves per record, zero on 4 waves per record.  Both produce
 * and consume the same garbled tables, so garbler and evaluator may differ; process-wide, takes
 * effect at the next launch.  Exists for A/B timing and for the interchangeability test. */
void lgc_set_split_kernels(int garbler, int evaluator);

/* The matrix-vector products of CGD at width 64 (src/cgd.oc:119-125, 96 % of the gates of a d = 500 solve) use a
 * Karatsuba multiplier -- three 32 x 32 arrays per product, 110 gate steps against 129 -- by default; 0 selects the
 * plain 64 x 64 array everywhere.  Same integers either way.  Process-wide, takes effect for programs built
 * afterwards; the two roles of one solve must agree (as on every other parameter of the program). */
void lgc_set_karatsuba(int on);
/* Co-located solvers created from now on: room in the table ring beyond the largest launch (what the garbler may run ahead of
 * the evaluator by), default 8 GiB, at most the largest launch again; 0 restores the default.  A block of a sharded sweep
 * needs little (its launches are few and large): eight ranks rehearsing an 8-GPU sweep on one MI355X set 512 MiB. */
void lgc_set_table_ring_slack(size_t bytes);

/* The gate hash of the half-gates scheme, H(x, t) = AES_k(sigma(x) ^ t) ^ sigma(x) ^ t under the fixed public key (this
 * library's counterpart of the gate hash inside Obliv-C's Yao runtime), on n labels (16 bytes each, tweaks[i]) on the
 * device: tests pin the kernels' hash to the CPU checker and to OpenSSL's AES with it. */
int lgc_gate_hash_eval(int device, const uint8_t *labels, const uint64_t *tweaks, uint8_t *out, size_t n);

/* Test hooks (tests/test_gpu_roles.py; not part of the drop-in surface).  lgc_test_party_garble_ring_stage
 * issues launch k as lgc_party_garble_ring does, in halves: stage 1 = the record kernel (stops before the table
 * pass of a critical-path launch; *is_critical_path tells whether the launch has one), stage 2 = the table pass.
 * lgc_test_party_ring_read copies `bytes` of the ring slot of launch k to the host -- what the mapped peer could
 * read at that moment. */
int lgc_test_party_garble_ring_stage(lgc_party *p, size_t launch, int stage, int *is_critical_path);
int lgc_test_party_ring_read(lgc_party *p, size_t launch, uint8_t *out, size_t bytes);

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
