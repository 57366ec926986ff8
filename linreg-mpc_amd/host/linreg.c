/* linreg.c -- `bin/linreg`: the reference's command line, roles, input file and stdout
 * contract (src/cmd/linreg.c:44-212), with every compute step on the MI355X through the C ABI
 * of liblinreg_gc (include/linreg_gc.h).  Host code stays C; this file only moves bytes.
 *
 *   linreg [Input_file] [Precision] [Party] [Algorithm] [Num. iterations CGD] [Lambda] [Options]
 *   Options: --use_ot, --prec_phase2=<p>, and (new; compile-time in the reference, Makefile:8-9)
 *            --width_phase1=<32|64>, --width_phase2=<32|64>
 *
 *   party 1 = CSP (trusted initializer in phase 1, garbler in phase 2)
 *   party 2 = Evaluator, parties >= 3 = data providers (README.md:41-45)
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <openssl/crypto.h>
#include <openssl/rand.h>
#include <poll.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "baseot.h"
#include "protocol.h"

/* ------------------------------------------------------------------------------ main */
typedef struct { size_t n, next; const uint32_t *launch; double *time; double t0; } iter_marks;
static void note_launch(size_t i, void *ctx) {
    iter_marks *m = ctx;
    if (i == 0) host_trace_mark("first table evaluated");
    while (m->next < m->n && m->launch[m->next] == i) m->time[m->next++] = wall_clock() - m->t0;
}

/* --devices: one block of the sweep per GPU, one thread and one table link per block */
enum { kMaxDevices = 16 };
typedef struct { table_link link; size_t lo, hi; int sending, rc; pthread_t th; } block_job;
static void *block_main(void *arg) {
    block_job *b = arg;
    b->rc = b->sending ? table_link_send_range(&b->link, b->lo, b->hi) : table_link_recv_range(&b->link, b->lo, b->hi, NULL, NULL);
    return NULL;
}
/* contiguous block [lo, hi) of block k out of K; sizes differ by at most one (python/sweep.py: partition) */
#define block_range sweep_block_range      /* host/sweep_plan.c */

/* The phase-2 object(s) of party 1 / 2: one, or with --devices one block of the sweep per entry (same seed: one set of
 * input labels, one label OT per data provider; block k starts at circuit lo_k, which keeps its gate ids disjoint), and for
 * the garbler its table ring(s).  A job, so that the CSP can run it on a thread while it is still the trusted initializer
 * of phase 1: lowering the program, the word file and -- above all -- a ring of several GB used to start after the barrier,
 * with every other party waiting. */
typedef struct {
    lgc_system sys; int role, device, n_devices, ring_slots; const int *devices; size_t table_chunk, n_lambdas; const double *lambdas;
    lgc_party **blocks, *party_obj; int rc; char err[256]; pthread_t th; int started;
} create_job;
static void *create_main(void *arg) {
    create_job *j = arg;
    uint8_t seed[16];
    const uint8_t *seedp = NULL;
    j->rc = 1;
    if (j->role == LGC_ROLE_GARBLER) {
        if (RAND_bytes(seed, sizeof seed) != 1) { snprintf(j->err, sizeof j->err, "RAND_bytes failed"); return NULL; }
        seedp = seed;
    }
#define JLGC(x) do { if ((x) != LGC_OK) { snprintf(j->err, sizeof j->err, "%s", lgc_last_error()); OPENSSL_cleanse(seed, sizeof seed); return NULL; } } while (0)
    if (j->n_devices) {
        for (int k = 0; k < j->n_devices; k++) {
            size_t lo, hi;
            block_range(j->n_lambdas, (size_t)j->n_devices, (size_t)k, &lo, &hi);
            JLGC(lgc_party_create_sweep_at(&j->blocks[k], j->devices[k], &j->sys, j->role, seedp, j->table_chunk, hi - lo, j->lambdas + lo, lo));
        }
        j->party_obj = j->blocks[0];
    } else if (j->n_lambdas) JLGC(lgc_party_create_sweep(&j->party_obj, j->device, &j->sys, j->role, seedp, j->table_chunk, j->n_lambdas, j->lambdas));
    else JLGC(lgc_party_create(&j->party_obj, j->device, &j->sys, j->role, seedp, j->table_chunk));
#undef JLGC
    OPENSSL_cleanse(seed, sizeof seed);
    /* (the table ring is NOT created here: hipIpcGetMemHandle on this thread, beside the initializer's own allocations and
     * exports on the main thread, failed with "invalid argument" in up to a third of the five-process runs on some boxes --
     * the main thread creates it after phase 1, create_rings below: a millisecond for a byte ring) */
    lgc_trace_mark(j->role == LGC_ROLE_GARBLER ? "garbler created" : "evaluator created");
    j->rc = 0;
    return NULL;
}
static int create_rings(create_job *j) {                            /* the garbler's ring(s), before the Evaluator asks for them */
    if (j->role != LGC_ROLE_GARBLER || j->ring_slots <= 0) return 0;
    for (int k = 0; k < (j->n_devices ? j->n_devices : 1); k++)
        if (tables_ring_prepare(j->n_devices ? j->blocks[k] : j->party_obj, j->ring_slots)) {
            snprintf(j->err, sizeof j->err, "could not create table ring %d", k);
            return 1;
        }
    return 0;
}
/* the HIP runtime and the device context come up on a thread of their own while main parses, connects and reads */
/* The CSP and the Evaluator live until the end of the protocol.  When one of them is gone -- its connection hung up -- the
 * other normally notices in its next recv() and leaves through check(); but it may sit in a call that never returns once the
 * peer is dead (seen: hipIpcOpenMemHandle on the ring of a CSP killed a moment earlier; a test of tests/test_host.py hit that
 * window once in six runs).  A watchdog thread polls the connection and ends the process with the reference's exit code for
 * every failure (src/cmd/linreg.c:206-211) -- but only for a hang-up that is a FAILURE: the peer left before this party had
 * everything it needs from it (g_peer_finished: set when the decode bits are in / out and the ring has been released), and
 * the main thread has not moved for five seconds since (host_progress: every launch and every trace mark ticks).  A healthy
 * CSP leaves before the Evaluator has printed its results, and POLLRDHUP fires on its FIN: a watchdog that only counted five
 * seconds from the hang-up would kill a slow but healthy Evaluator (a shared GPU, a profiler, a blocked stdout). */
static volatile int g_protocol_over;      /* 1: results are out, cleaning up; 2: main is about to return */
static volatile int g_peer_finished;      /* nothing more is needed from the watched peer: its hang-up is not an event */
typedef struct { int fd, party, peer; } watchdog_arg;
static void *peer_watchdog(void *arg) {
    watchdog_arg *w = arg;
    struct pollfd pf = {w->fd, POLLRDHUP, 0};
    for (;;) {
        if (g_protocol_over >= 2 || g_peer_finished) return NULL;
        pf.revents = 0;
        int r = poll(&pf, 1, 250);
        if (r > 0 && (pf.revents & (POLLRDHUP | POLLHUP | POLLERR | POLLNVAL))) break;
    }
    unsigned long seen = host_progress();
    for (int quiet = 0; quiet < 20;) {                               /* 20 x 250 ms without a sign of life */
        if (g_protocol_over >= 2 || g_peer_finished) return NULL;
        usleep(250000);
        unsigned long now = host_progress();
        if (now != seen) { seen = now; quiet = 0; } else quiet++;
    }
    if (g_protocol_over >= 2 || g_peer_finished) return NULL;
    if (g_protocol_over == 1) { fflush(stdout); _exit(0); }          /* the protocol was through: only the clean-up is stuck */
    fprintf(stderr, "party %d: party %d is gone and this party is stuck in a call that does not return; giving up\n", w->party, w->peer);
    _exit(1);
}

static int g_warm_what;
static void *warm_main(void *arg) {
    int device = *(int *)arg;
    if (lgc_device_warm(device) == LGC_OK && g_warm_what) (void)lgc_preload(device, g_warm_what);
    return NULL;
}

/* One thread per data provider on the CSP: the base OTs (128 P-256 transfers, host work on both sides) and then the label
 * OT of that provider's share (dcrRecvBitArray, src/input.c:94-108) -- its own connection, its own OpenSSL objects, its own
 * OT session and page-locked buffers.  The providers are independent of each other; served one after the other, as in
 * rounds 1-3, the label OTs were 20-25 ms of latency EACH on the critical path of every run. */
typedef struct {
    node *self; int peer, device, rc, ring; lgc_party *po; size_t share, bits; char err[256]; pthread_t th;
} input_ot_job;
/* --input_ring (every party on this node): the messages of the label OT stay in HBM.  The provider owns a device buffer
 * [u | e] that the CSP maps (hipIpc): it holds what the socket would carry between those two -- the provider's masked columns u,
 * the CSP's ciphertexts e -- and nothing else; the label PAIRS stay in two device buffers private to the CSP; the chosen labels
 * go to the Evaluator through a second buffer of the provider's that only the Evaluator maps.  Sockets carry the 64-byte
 * handles and one-byte tokens.  (config 4: 256 MB per provider that used to cross a socket and PCIe twice.) */
static int input_ot_ring_csp(input_ot_job *j, lgc_ot_sender *S) {
    const size_t bits = j->bits, ub = lgc_ot_u_bytes(bits);
    void *d0 = NULL, *d1 = NULL, *ue = NULL;
    uint8_t h[64], tok = 1;
    int rc = 1;
    if (lgc_ot_sender_set_device_io(S, 1) != LGC_OK) goto out;
    if (lgc_dev_alloc(j->device, bits * 16, &d0, NULL) != LGC_OK || lgc_dev_alloc(j->device, bits * 16, &d1, NULL) != LGC_OK) goto out;
    if (lgc_party_input_pairs_dev(j->po, j->share, d0, d1) != LGC_OK) goto out;
    lgc_trace_mark("input OT: label pairs exported");
    if (recv_blob(j->self, j->peer, h, sizeof h)) { snprintf(j->err, sizeof j->err, "OT: no buffer handle from party %d", j->peer); goto out2; }
    if (lgc_dev_open(j->device, h, &ue) != LGC_OK) goto out;
    lgc_trace_mark("input OT: u received");
    if (lgc_ot_labels_send(S, d0, d1, bits, ue, (uint8_t *)ue + ub) != LGC_OK) goto out;
    lgc_trace_mark("input OT: ciphertexts computed");
    lgc_dev_close(ue); ue = NULL;
    if (send_blob(j->self, j->peer, &tok, 1)) { snprintf(j->err, sizeof j->err, "OT: could not signal party %d", j->peer); goto out2; }
    rc = 0;
    goto out2;
out:
    snprintf(j->err, sizeof j->err, "%s", lgc_last_error());
out2:
    if (ue) lgc_dev_close(ue);
    lgc_dev_free_secret(d0, bits * 16);                              /* both labels of every input bit: their XOR is R */
    lgc_dev_free_secret(d1, bits * 16);
    return rc;
}
static void *input_ot_main(void *arg) {
    input_ot_job *j = arg;
    uint8_t delta[16], seeds[128][16];
    uint8_t *m0 = NULL, *m1 = NULL, *u = NULL, *e = NULL;
    lgc_ot_sender *S = NULL;
    const size_t bits = j->bits;
    j->rc = 1;
#define JFAIL(...) do { snprintf(j->err, sizeof j->err, __VA_ARGS__); goto out; } while (0)
    if (baseot_ext_sender(j->self, j->peer, delta, seeds)) JFAIL("base OT with party %d failed", j->peer);
    lgc_trace_mark("base OTs done");
    if (j->ring) {
        if (lgc_ot_sender_create(&S, j->device, delta, seeds) != LGC_OK) JFAIL("%s", lgc_last_error());
        if (input_ot_ring_csp(j, S)) goto out;
        j->rc = 0;
        goto out;
    }
    m0 = lgc_host_alloc(bits * 16); m1 = lgc_host_alloc(bits * 16); u = lgc_host_alloc(lgc_ot_u_bytes(bits)); e = lgc_host_alloc(bits * 32);
    if (!m0 || !m1 || !u || !e) JFAIL("%s", lgc_last_error());
    lgc_trace_mark("input OT: page-locked buffers");
    if (lgc_ot_sender_create(&S, j->device, delta, seeds) != LGC_OK) JFAIL("%s", lgc_last_error());
    lgc_trace_mark("input OT: sender session");
    if (lgc_party_input_pairs(j->po, j->share, m0, m1) != LGC_OK) JFAIL("%s", lgc_last_error());
    lgc_trace_mark("input OT: label pairs exported");
    if (recv_blob(j->self, j->peer, u, lgc_ot_u_bytes(bits))) JFAIL("OT: could not receive u from party %d", j->peer);
    lgc_trace_mark("input OT: u received");
    if (lgc_ot_labels_send(S, m0, m1, bits, u, e) != LGC_OK) JFAIL("%s", lgc_last_error());
    lgc_trace_mark("input OT: ciphertexts computed");
    if (send_blob(j->self, j->peer, e, bits * 32)) JFAIL("OT: could not send to party %d", j->peer);
    j->rc = 0;
out:
#undef JFAIL
    if (S) lgc_ot_sender_destroy(S);
    if (m0) OPENSSL_cleanse(m0, bits * 16);                          /* both labels of every input bit: their XOR is R */
    if (m1) OPENSSL_cleanse(m1, bits * 16);
    lgc_host_free(m0); lgc_host_free(m1); lgc_host_free(u); lgc_host_free(e);
    OPENSSL_cleanse(delta, sizeof delta); OPENSSL_cleanse(seeds, sizeof seeds);   /* base-OT delta and seeds */
    return NULL;
}

int main(int argc, char **argv) {
    uint64_t *share_A = NULL, *share_b = NULL;
    config *c = NULL;
    node *self = NULL;
    lgc_party *party_obj = NULL;
    lgc_party *blocks[kMaxDevices] = {0};       /* --devices: blocks[0] == party_obj */
    int devices[kMaxDevices], n_devices = 0;
    int status;
    create_job cj;
    memset(&cj, 0, sizeof cj);

    check(argc > 6, "Usage: %s [Input_file] [Precision] [Party] [Algorithm] [Num. iterations CGD] [Lambda] [Options]\n"
          "Options: --use_ot: Enables the OT-based phase 1 protocol\n"
          "         --prec_phase2=<Precision phase 2>: Use different precision for phase 2 of the protocol\n"
          "         --width_phase1=<32|64>, --width_phase2=<32|64>: bit widths (default 64)\n"
          "         --table_ring[=slots]: parties 1 and 2 share one node; garbled tables stay in HBM (a byte ring of the largest\n"
          "                  launch plus slack; =slots: that many slots of the largest launch instead)\n"
          "         --table_lanes=<K>: garbled tables through the network over K extra TCP connections\n"
          "         --ti_ring: (TI mode) all parties on this node: the vectors of the multiplication protocol stay in HBM\n"
          "         --ot_ring: --use_ot with all data providers on this node: the OT extension's messages stay in HBM\n"
          "         --input_ring: (every party, all on this node) the messages of the label OT of phase 2 stay in HBM\n"
          "         --lambdas=l1,l2,...: regularisation sweep -- one circuit per value on the same shares (the data\n"
          "                  providers share their inputs once); [Lambda] is then ignored\n"
          "         --devices=g0,g1,...: (parties 1 and 2, with --lambdas and --table_ring) contiguous blocks of the sweep on\n"
          "                  these GPUs, one block per entry (an index may repeat); without it LINREG_DEVICE (default 0)", argv[0]);
    char *end;
    errno = 0;
    int precision = (int)strtol(argv[2], &end, 10);
    check(!errno, "strtol: %s", strerror(errno));
    check(!*end, "Precision must be a number");
    int party = (int)strtol(argv[3], &end, 10);
    check(!errno, "strtol: %s", strerror(errno));
    check(!*end, "Party must be a number");
    { char tag_[16]; snprintf(tag_, sizeof tag_, "p%d", party); lgc_trace_set_tag(tag_); }
    lgc_trace_mark("main entered");
    char *algorithm = argv[4];
    check(!strcmp(algorithm, "cholesky") || !strcmp(algorithm, "ldlt") || !strcmp(algorithm, "cgd"),
          "Algorithm must be cholesky, ldlt, or cgd.");
    double lambda = strtod(argv[6], &end);
    check(!errno, "strtod: %s", strerror(errno));
    check(!*end, "lambda must be a number");

    int use_ot = 0, precision_phase2 = -1, w1 = 64, w2 = 64, ring_slots = 0, table_lanes = 0, input_ring = 0;
    double *lambdas = NULL;                     /* --lambdas: the per-lambda sweep (lambda enters at linear.oc:52-57) */
    size_t n_lambdas = 0;
    for (int i = 7; i < argc; i++) {
        if (!strcmp(argv[i], "--use_ot")) use_ot |= 1;
        else if (!strcmp(argv[i], "--ot_ring")) use_ot |= 3;
        else if (!strcmp(argv[i], "--input_ring")) input_ring = 1;         /* all parties on this node: the label OT's messages stay in HBM */
        else if (!strcmp(argv[i], "--ti_ring")) protocol_set_ti_ring(1);   /* TI mode, all parties on this node: vectors stay in HBM */      /* --use_ot with u / y of the extension in device rings */
        else if (!strncmp(argv[i], "--lambdas=", 10)) {
            const char *q = argv[i] + 10;
            while (*q) {
                char *e2;
                double v = strtod(q, &e2);
                check(e2 != q && (*e2 == ',' || !*e2), "--lambdas wants a comma-separated list of numbers");
                lambdas = realloc(lambdas, (n_lambdas + 1) * sizeof *lambdas);
                lambdas[n_lambdas++] = v;
                q = *e2 ? e2 + 1 : e2;
            }
            check(n_lambdas > 0, "--lambdas wants at least one value");
        }
        else if (!strncmp(argv[i], "--devices=", 10)) {
            n_devices = sweep_parse_devices(argv[i] + 10, devices, kMaxDevices);
            check(n_devices > 0, "--devices wants a comma-separated list of at most %d device indices", kMaxDevices);
        }
        else if (!strcmp(argv[i], "--table_ring")) ring_slots = TABLE_RING_BYTES;
        else if (sscanf(argv[i], "--table_ring=%i", &ring_slots) == 1) {}
        else if (sscanf(argv[i], "--table_lanes=%i", &table_lanes) == 1) protocol_set_table_lanes(table_lanes);
        else if (sscanf(argv[i], "--width_phase1=%i", &w1) == 1) {}
        else if (sscanf(argv[i], "--width_phase2=%i", &w2) == 1) {}
        else if (sscanf(argv[i], "--prec_phase2=%i", &precision_phase2) != 1) precision_phase2 = -1;
    }
    if (precision_phase2 != -1) printf("Using different precision for phase 2 (%i)\n", precision_phase2);
    check((w1 == 32 || w1 == 64) && (w2 == 32 || w2 == 64), "Bit widths must be 32 or 64");
    check(precision >= 0, "Precision of phase 1 must be nonnegative");
    check(precision_phase2 >= -1, "Precision of phase 2 must be nonnegative");
    check(precision < w1, "Precision of phase 1 must be smaller than bit size of phase 1");
    check(precision_phase2 < w2, "Precision of phase 2 must be smaller than bit size of phase 2");
    int num_iterations = !strcmp(algorithm, "cgd") ? atoi(argv[5]) : 0;
    int device = getenv("LINREG_DEVICE") ? atoi(getenv("LINREG_DEVICE")) : 0;
    if (n_devices) {
        check(n_lambdas > 0 && ring_slots > 0, "--devices shards a --lambdas sweep and needs --table_ring (CSP and Evaluator on this node)");
        if ((size_t)n_devices > n_lambdas) n_devices = (int)n_lambdas;       /* no empty blocks */
        if (party <= 2) {
            /* before anything is allocated: every index exists, and distinct devices can reach each other (the shared prefix
             * travels by peer copy, lgc_party_share_prefix; the Evaluator maps the CSP's table rings over hipIpc) */
            check(lgc_devices_preflight(devices, (size_t)n_devices) == LGC_OK, "--devices: %s", lgc_last_error());
            device = devices[0];
        }
    }

    /* HIP runtime + device context (60-150 ms with four or five processes starting at once): on a thread, beside the
     * configuration file, the socket mesh and a data provider's parsing of its columns */
    static int warm_device;
    pthread_t warm_th;
    warm_device = device;
    /* the code objects this party will launch from, and streams for its OT sessions: data providers run phase 1 and the label
     * OT, the CSP is the initializer (phase-1 kernels) and the label-OT sender; the Evaluator's kernels come with its program */
    g_warm_what = party == 2 ? 0 : 3;   /* (no pooled streams: every extra hardware queue costs ~60 ms when four processes tear down at once) */
    int warm_started = pthread_create(&warm_th, NULL, warm_main, &warm_device) == 0;
    if (warm_started) pthread_detach(warm_th);

    status = config_new(&c, argv[1]);
    check(!status, "Could not read config");
    c->party = party;
    check(party >= 1 && party <= c->num_parties, "Party must be in 1..%d", c->num_parties);

    lgc_trace_mark("configuration read");
    double time = wall_clock();
    /* LINREG_TRACE=1: wall-clock marks on stderr (where an end-to-end run of a small configuration spends its time) */
    /* (LGCT lines on the system-wide monotonic clock, shared with the library's own marks: lgc_trace_mark) */
#define TRACE(what) host_trace_mark(what)
    if (party == 2) printf("{\"n\":\"%zd\", \"d\":\"%zd\" \"p\":\"%d\"}\n", c->n, c->d, c->num_parties - 1);

    status = node_new(&self, party, c->num_parties, c->endpoint);
    check(!status, "Could not create node");
    TRACE("connected");
    static watchdog_arg wd;
    if (party <= 2) {                                                /* parties 1 and 2 watch each other's connection */
        pthread_t wt;
        wd.fd = self->fd[(3 - party) - 1]; wd.party = party; wd.peer = 3 - party;
        if (wd.fd >= 0 && !pthread_create(&wt, NULL, peer_watchdog, &wd)) pthread_detach(wt);
    }

    /* the phase-2 system: known from the configuration before any protocol message */
    const int precision2 = precision_phase2 != -1 ? precision_phase2 : precision;
    const size_t d = c->d, T = d * (d + 1) / 2;
    const int P = c->num_parties - 2;
    lgc_system sys;
    memset(&sys, 0, sizeof sys);
    sys.d = d; sys.width = w2; sys.precision = precision2;
    sys.algorithm = !strcmp(algorithm, "cholesky") ? LGC_ALG_CHOLESKY : (!strcmp(algorithm, "ldlt") ? LGC_ALG_LDLT : LGC_ALG_CGD);
    sys.num_iterations = num_iterations; sys.lambda = lambda; sys.nshares = (size_t)P;
    sys.normalize = 1; sys.reveal_inputs = 1; sys.trace = 1;
    if (n_lambdas) { sys.reveal_inputs = 0; sys.trace = 0; }       /* merged program of n_lambdas circuits: results only */
    /* table bytes per launch: socket mode moves them through host buffers; ring mode keeps them in HBM (CSP and Evaluator on
     * one node), so launches are as large as the fused solver's: 2^25 gate steps = 64 GiB, i.e. a whole d = 500 matrix-vector
     * product is ONE launch.  Rounds 2-4 cut at 16 GiB: the product then went out as seven launches of 4.4 rounds of the
     * chip each, the garbler's launch k + 1 waited for the evaluation of launch k (4 GiB of run-ahead room), and the
     * two-process solve ran 7-9 % behind the co-located one -- profiles/r5_timeline_d500_two_process.txt, and
     * scripts/exp/two_proc_shape_ab.sh: 1.94 s at 16 GiB, 1.87 s at 32 GiB, 1.82 s at 64 GiB (co-located: 1.81 s) */
    const size_t kTableChunk = ring_slots > 0 ? (size_t)64 << 30 : (size_t)64 << 20;
    cj.sys = sys; cj.device = device; cj.n_devices = n_devices; cj.devices = devices; cj.ring_slots = ring_slots;
    cj.table_chunk = kTableChunk; cj.n_lambdas = n_lambdas; cj.lambdas = lambdas; cj.blocks = blocks;
    cj.role = party == 1 ? LGC_ROLE_GARBLER : LGC_ROLE_EVALUATOR;

    if (party == 1) {
        /* the garbler of phase 2 is built while this process is the trusted initializer (TI mode) or idle (OT mode) */
        check(!pthread_create(&cj.th, NULL, create_main, &cj), "could not start the garbler's creation thread");
        cj.started = 1;
        if (!use_ot) {
            status = run_trusted_initializer(self, c, w1, device);
            check(!status, "Error while running trusted initializer");
        }
    } else if (party > 2) {
        status = run_party(self, c, precision, precision_phase2 != -1 ? precision_phase2 : precision, w1, w2, use_ot, device,
                           &share_A, &share_b);
        check(!status, "Error while running party %d", party);
    } else {
        /* The Evaluator has no part in phase 1: it brings up its GPU context, program and buffers while the data
         * providers work, instead of after the barrier with the CSP waiting for it (0.3 s of a 0.9 s config-3 run). */
        create_main(&cj);
        check(!cj.rc, "%s", cj.err);
        party_obj = cj.party_obj;
    }
    TRACE("phase 1 done");
    check(!net_barrier(self), "Error while waiting for other peers to finish");
    TRACE("barrier");
    printf("Party %d finished phase 1\n", party);

    /* ---------------------------------------------------------------------------- phase 2 */
    if (precision_phase2 != -1) precision = precision_phase2;
    if (party == 1) {                                                /* CSP: garbler */
        pthread_join(cj.th, NULL);
        cj.started = 0;
        check(!cj.rc, "%s", cj.err);
        party_obj = cj.party_obj;
        check(!create_rings(&cj), "%s", cj.err);
        if (!n_devices) {                                            /* the Evaluator maps the ring during the label OT */
            check(!programs_agree(self, 2, party_obj, 1), "program check failed");
            check(!tables_ring_meet(self, 2, party_obj, 1, ring_slots), "could not offer the table ring");
        }
        input_ot_job *jobs = calloc((size_t)P, sizeof *jobs);
        check(jobs != NULL, "out of memory");
        int started_ot = 0, bad_ot = 0;
        for (int k = 3; k <= c->num_parties; k++) {                  /* data providers: share k - 3 (linear.oc:31) */
            input_ot_job *j = &jobs[k - 3];
            j->self = self; j->peer = k; j->device = device; j->po = party_obj; j->share = (size_t)(k - 3);
            j->bits = lgc_party_input_bits(party_obj);
            j->ring = input_ring;
            if (pthread_create(&j->th, NULL, input_ot_main, j)) break;
            started_ot++;
        }
        for (int k = 0; k < started_ot; k++) {
            pthread_join(jobs[k].th, NULL);
            if (jobs[k].rc) { fprintf(stderr, "%s\n", jobs[k].err); bad_ot = 1; }
        }
        bad_ot |= started_ot != P;
        free(jobs);
        check(!bad_ot, "input sharing with the data providers failed");
        TRACE("input labels sent");
        if (n_devices) {
            /* one table link (hipIpc ring + token connection) and one thread per block.  Block 0 garbles the shared prefix
             * first; its share sums go to the other GPUs over xGMI; then all blocks run side by side */
            int fds[kMaxDevices];
            block_job jb[kMaxDevices];
            memset(jb, 0, sizeof jb);
            for (int k = 0; k < n_devices; k++) check(!programs_agree(self, 2, blocks[k], 1), "program check failed (block %d)", k);
            check(!net_lanes_offer(self, 2, n_devices, fds), "could not open %d token connections to the Evaluator", n_devices);
            const size_t pre = lgc_party_prefix_launches(blocks[0]);
            for (int k = 0; k < n_devices; k++)
                check(!table_link_open(&jb[k].link, self, 2, fds[k], blocks[k], 1, ring_slots, k ? pre : 0), "could not open table link %d", k);
            check(!table_link_send_range(&jb[0].link, 0, pre), "could not stream the prefix tables");
            for (int k = 1; k < n_devices; k++) LGC(lgc_party_share_prefix(blocks[k], blocks[0]));
            TRACE("prefix garbled and shared");
            for (int k = 0; k < n_devices; k++) {
                jb[k].lo = pre; jb[k].hi = lgc_party_num_launches(blocks[k]); jb[k].sending = 1;
                check(!pthread_create(&jb[k].th, NULL, block_main, &jb[k]), "could not start a block thread");
            }
            int bad = 0;
            for (int k = 0; k < n_devices; k++) { pthread_join(jb[k].th, NULL); bad |= jb[k].rc; }
            check(!bad, "could not stream the garbled tables");
            TRACE("tables sent");
            for (int k = 0; k < n_devices; k++) {
                size_t nr = lgc_party_num_reveal(blocks[k]);
                uint64_t *dec = malloc((nr + 1) * 8);
                LGC(lgc_party_decode_bits(blocks[k], dec));
                check(!send_blob(self, 2, dec, nr * 8), "could not send decode bits");
                free(dec);
            }
            /* the rings stay until the Evaluator has evaluated every launch of every block (table_link_finish) */
            for (int k = 0; k < n_devices; k++) check(!table_link_finish(&jb[k].link, 1), "the Evaluator did not release table ring %d", k);
            g_peer_finished = 1;
            for (int k = 0; k < n_devices; k++) close(fds[k]);
            goto done;
        }
        check(!tables_send(self, 2, party_obj, ring_slots, kTableChunk), "could not stream the garbled tables");
        TRACE("tables sent");
        size_t nr = lgc_party_num_reveal(party_obj);
        uint64_t *dec = malloc((nr + 1) * 8);
        LGC(lgc_party_decode_bits(party_obj, dec));
        TRACE("decode bits read");
        check(!send_blob(self, 2, dec, nr * 8), "could not send decode bits");
        free(dec);
        g_peer_finished = 1;                                     /* (ring mode: tables_send returned with the ring released) */
    } else if (party == 2) {                                         /* Evaluator */
        double time_start = wall_clock();
        printf("\nAlgorithm: %s\n", algorithm);
        size_t bits = lgc_party_input_bits(party_obj);
        uint8_t *labels = malloc(bits * 16);
        if (!n_devices) {                                         /* (see the CSP's side above) */
            check(!programs_agree(self, 1, party_obj, 0), "program check failed");
            check(!tables_ring_meet(self, 1, party_obj, 0, ring_slots), "could not map the table ring");
        }
        printf("party %d listening for %d inputs", party, P);
        for (int k = 3; k <= c->num_parties; k++) {
            if (input_ring) {                                     /* the provider's label buffer, mapped; one byte back when it may go */
                uint8_t h[64], tok = 1;
                void *dl = NULL;
                check(!recv_blob(self, k, h, sizeof h), "could not receive the label buffer of party %d", k);
                LGC(lgc_dev_open(device, h, &dl));
                int rc_ = lgc_party_set_input_labels_dev(party_obj, (size_t)(k - 3), dl);
                lgc_dev_close(dl);
                check(rc_ == LGC_OK, "%s", lgc_last_error());
                check(!send_blob(self, k, &tok, 1), "could not release party %d", k);
                printf("Evaluator received A from party %d\nEvaluator received b from party %d\n", k, k);
                continue;
            }
            check(!recv_blob(self, k, labels, bits * 16), "could not receive labels from party %d", k);
            LGC(lgc_party_set_input_labels(party_obj, (size_t)(k - 3), labels));
            printf("Evaluator received A from party %d\nEvaluator received b from party %d\n", k, k);
        }
        free(labels);
        TRACE("input labels received");
        double t_ot = wall_clock() - time_start;
        /* where cgd.oc:190-194 prints yaoGateCount() and the running time: after the launch that
         * completes each iteration */
        size_t n_marks = (sys.algorithm == LGC_ALG_CGD && !n_lambdas) ? (size_t)num_iterations : 0;
        uint32_t *mark_launch = malloc((n_marks + 1) * sizeof *mark_launch);
        uint64_t *mark_gates = malloc((n_marks + 1) * sizeof *mark_gates);
        double *mark_time = malloc((n_marks + 1) * sizeof *mark_time);
        if (n_marks) LGC(lgc_party_iteration_marks(party_obj, mark_launch, mark_gates, n_marks));
        iter_marks marks = {n_marks, 0, mark_launch, mark_time, time_start};
        int64_t *beta = malloc((n_lambdas ? n_lambdas : 1) * d * 8), *ab = malloc((T + d) * 8),
                *trace = malloc(((size_t)num_iterations * (d + 4) + 1) * 8);
        unsigned long long total_gates = 0;
        if (n_devices) {                                          /* the CSP's counterpart, block by block */
            int fds[kMaxDevices], got = 0;
            block_job jb[kMaxDevices];
            memset(jb, 0, sizeof jb);
            for (int k = 0; k < n_devices; k++) check(!programs_agree(self, 1, blocks[k], 0), "program check failed (block %d)", k);
            check(!net_lanes_accept_offer(self, 1, kMaxDevices, &got, fds) && got == n_devices, "the CSP offered %d table links, expected %d (same --devices on both?)", got, n_devices);
            const size_t pre = lgc_party_prefix_launches(blocks[0]);
            for (int k = 0; k < n_devices; k++)
                check(!table_link_open(&jb[k].link, self, 1, fds[k], blocks[k], 0, ring_slots, k ? pre : 0), "could not open table link %d", k);
            check(!table_link_recv_range(&jb[0].link, 0, pre, NULL, NULL), "could not evaluate the prefix");
            for (int k = 1; k < n_devices; k++) LGC(lgc_party_share_prefix(blocks[k], blocks[0]));
            TRACE("prefix evaluated and shared");
            for (int k = 0; k < n_devices; k++) {
                jb[k].lo = pre; jb[k].hi = lgc_party_num_launches(blocks[k]); jb[k].sending = 0;
                check(!pthread_create(&jb[k].th, NULL, block_main, &jb[k]), "could not start a block thread");
            }
            int bad = 0;
            for (int k = 0; k < n_devices; k++) { pthread_join(jb[k].th, NULL); bad |= jb[k].rc; }
            check(!bad, "could not receive garbled tables");
            TRACE("tables evaluated");
            for (int k = 0; k < n_devices; k++) check(!table_link_finish(&jb[k].link, 0), "could not release table ring %d", k);
            for (int k = 0; k < n_devices; k++) {
                size_t lo, hi, nr = lgc_party_num_reveal(blocks[k]);
                block_range(n_lambdas, (size_t)n_devices, (size_t)k, &lo, &hi);
                uint64_t *dec = malloc((nr + 1) * 8);
                check(!recv_blob(self, 1, dec, nr * 8), "could not receive decode bits");
                LGC(lgc_party_finish(blocks[k], dec, beta + lo * d, NULL, NULL));
                free(dec);
                /* every block's program holds the prefix; it was garbled once */
                total_gates += lgc_party_and_gates(blocks[k]) - (k ? lgc_party_prefix_and_gates(blocks[k]) : 0);
                close(fds[k]);
            }
            g_peer_finished = 1;                                  /* every block's decode bits are in: the CSP may go */
        } else {
            check(!tables_recv(self, 1, party_obj, ring_slots, kTableChunk, note_launch, &marks), "could not receive garbled tables");
            TRACE("tables evaluated");
            size_t nr = lgc_party_num_reveal(party_obj);
            uint64_t *dec = malloc((nr + 1) * 8);
            check(!recv_blob(self, 1, dec, nr * 8), "could not receive decode bits");
            g_peer_finished = 1;                                  /* the CSP may go: nothing more comes from it */
            LGC(lgc_party_finish(party_obj, dec, beta, n_lambdas ? NULL : trace, n_lambdas ? NULL : ab));
            free(dec);
            total_gates = lgc_party_and_gates(party_obj);
        }
        if (n_lambdas) {                                         /* sweep: one Result line per lambda, in order */
            printf("Time taken for OT: %f\nOT time: %f\n", t_ot, t_ot);
            printf("Time elapsed: %f\n", wall_clock() - time);
            printf("Number of gates: %llu\n", total_gates);
            for (size_t t = 0; t < n_lambdas; t++) {
                printf("Lambda: %.17g\nResult: ", lambdas[t]);
                for (size_t i = 0; i < d; i++) printf("%20.15f ", fixed_to_double(beta[t * d + i], precision));
                printf("\n");
            }
            free(beta); free(ab); free(trace); free(mark_launch); free(mark_gates); free(mark_time);
            goto done;
        }
        /* debug reveal of A and b (src/linear.oc:68-88) */
        printf("A = \n");
        for (size_t i = 0; i < d; i++) {
            for (size_t j = 0; j <= i; j++) printf("%3.8f ", fixed_to_double(ab[idx(i, j)], precision));
            printf("\n");
        }
        printf("b = \n");
        for (size_t i = 0; i < d; i++) printf("%3.6f ", fixed_to_double(ab[T + i], precision));
        printf("\n");
        printf("Time taken for OT: %f\n", t_ot);
        if (sys.algorithm == LGC_ALG_CGD) {
            printf("OT time: %f\nStarting iterations.\n", t_ot);
            for (int t = 0; t < num_iterations; t++) {                /* src/cgd.oc:167-194 */
                const int64_t *row = trace + (size_t)t * (d + 4);
                printf("Iteration %d (x):\n", t);
                for (size_t i = 0; i < d; i++) printf("%20.15f ", fixed_to_double(row[i], precision));
                printf("\nGamma: %30.20f ", fixed_to_double(row[d], precision));
                printf("\nEta: %30.20f ", fixed_to_double(row[d + 1], precision));
                printf("\nq: %30.20f ", fixed_to_double(row[d + 2], precision));
                printf("\nng: %30.20f ", fixed_to_double(row[d + 3], precision));
                printf("\nIteration %d gate count: %llu", t, (unsigned long long)mark_gates[t]);
                printf("\nIteration %d time: %f\n", t, mark_time[t]);
            }
        } else {
            printf("OT time: %f\n", t_ot);
        }
        printf("Time elapsed: %f\n", wall_clock() - time);                                   /* linreg.c:182 */
        printf("Number of gates: %lld\n", (long long)lgc_party_and_gates(party_obj));         /* linreg.c:183 */
        printf("Result: ");                                                                 /* linreg.c:184-187 */
        for (size_t i = 0; i < d; i++) printf("%20.15f ", fixed_to_double(beta[i], precision));
        printf("\n");
        free(beta); free(ab); free(trace); free(mark_launch); free(mark_gates); free(mark_time);
    } else {                                                         /* data provider (linreg.c:192-198, input.c:23-50) */
        printf("party %d connecting to CSP and Evaluator\n", party);
        uint8_t s0[128][16], s1[128][16];
        check(!baseot_ext_receiver(self, 1, s0, s1), "base OT with the CSP failed");
        TRACE("base OT done");
        printf("party %d connected successfully to CSP and Evaluator\n", party);
        lgc_ot_receiver *R = 0;
        LGC(lgc_ot_receiver_create(&R, device, s0, s1));
        TRACE("input OT: receiver session");
        const size_t words = T + d, bits = words * (size_t)w2;
        if (input_ring) {                                            /* see input_ot_ring_csp */
            const size_t ub = lgc_ot_u_bytes(bits);
            uint8_t *selh = malloc(bits), hue[64], hl[64], tok = 0;
            void *dsel = NULL, *due = NULL, *dl = NULL;
            for (size_t i = 0; i < words; i++) {
                uint64_t v = i < T ? share_A[i] : share_b[i - T];
                for (int j = 0; j < w2; j++) selh[i * (size_t)w2 + (size_t)j] = (uint8_t)((v >> j) & 1);
            }
            LGC(lgc_ot_receiver_set_device_io(R, 1));
            LGC(lgc_dev_alloc(device, bits, &dsel, NULL));
            LGC(lgc_dev_alloc(device, ub + bits * 32, &due, hue));
            LGC(lgc_dev_alloc(device, bits * 16, &dl, hl));
            LGC(lgc_dev_upload(dsel, selh, bits));
            OPENSSL_cleanse(selh, bits); free(selh);
            LGC(lgc_ot_labels_recv_start(R, dsel, bits, due));
            check(!send_blob(self, 1, hue, sizeof hue), "OT: could not hand the buffer to the CSP");
            TRACE("input OT: u sent");
            check(!recv_blob(self, 1, &tok, 1), "OT: the CSP did not answer");
            TRACE("input OT: ciphertexts received");
            LGC(lgc_ot_labels_recv_finish(R, (uint8_t *)due + ub, dl));
            check(!send_blob(self, 2, hl, sizeof hl), "could not hand the labels to the Evaluator");   /* input.c:46 */
            check(!recv_blob(self, 2, &tok, 1), "the Evaluator did not take the labels");
            TRACE("labels forwarded to the Evaluator");
            lgc_ot_receiver_destroy(R);
            lgc_dev_free_secret(dsel, bits); lgc_dev_free(due); lgc_dev_free_secret(dl, bits * 16);
            goto done;
        }
        uint8_t *sel = malloc(bits), *u = malloc(lgc_ot_u_bytes(bits)), *e = malloc(bits * 32), *labels = malloc(bits * 16);
        for (size_t i = 0; i < words; i++) {                         /* sel[i*intsize+j] = (input[i]>>j)&1 (input.c:41) */
            uint64_t v = i < T ? share_A[i] : share_b[i - T];
            for (int j = 0; j < w2; j++) sel[i * (size_t)w2 + (size_t)j] = (uint8_t)((v >> j) & 1);
        }
        LGC(lgc_ot_labels_recv_start(R, sel, bits, u));
        check(!send_blob(self, 1, u, lgc_ot_u_bytes(bits)), "OT: could not send u to the CSP");
        TRACE("input OT: u sent");
        check(!recv_blob(self, 1, e, bits * 32), "OT: could not receive from the CSP");
        TRACE("input OT: ciphertexts received");
        LGC(lgc_ot_labels_recv_finish(R, e, labels));
        check(!send_blob(self, 2, labels, bits * 16), "could not forward labels to the Evaluator");   /* input.c:46 */
        TRACE("labels forwarded to the Evaluator");
        lgc_ot_receiver_destroy(R);
        free(sel); free(u); free(e); free(labels);
    }

done:
    g_protocol_over = 1;
    TRACE("protocol done");
    for (int k = 1; k < n_devices; k++) if (blocks[k]) lgc_party_destroy(blocks[k]);
    if (party_obj) lgc_party_destroy(party_obj);           /* (wipes label material before its memory is released) */
    node_destroy(&self);
    config_destroy(&c);
    free(share_A);
    free(share_b);
    free(lambdas);
    g_protocol_over = 2;
    TRACE("exit");
    /* (leaving through _exit() to skip the HIP runtime's exit handlers -- 70-80 ms per process -- was measured and is WORSE:
     * the kernel driver then tears the process's queues and mappings down by itself, 250 ms for config 2) */
    return 0;
error:
    if (cj.started) { pthread_join(cj.th, NULL); if (!party_obj) party_obj = cj.party_obj; }
    for (int k = 1; k < n_devices; k++) if (blocks[k]) lgc_party_destroy(blocks[k]);
    if (party_obj) lgc_party_destroy(party_obj);
    config_destroy(&c);
    node_destroy(&self);
    free(share_A);
    free(share_b);
    g_protocol_over = 2;
    return 1;
}
