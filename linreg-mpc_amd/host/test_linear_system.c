/* test_linear_system.c -- two-party phase-2 benchmark with the reference's command line
 * (src/cmd/test/test_linear_system.c:81-137):
 *     test_linear_system [Port] [Party] [Input file] [Algorithm] [Num. iterations CGD] [Precision]
 * Both parties read the same clear system (A, b); the "shares" are fabricated with the constant
 * masks 123456 / 0xDEADBEEF (:34-44); party 1 (server, garbler) holds the masks, party 2
 * (evaluator) the masked values; the circuit adds them (src/linear.oc:96-135).  Party 2 prints the
 * lines experiments/test_phase2_aws.py parses.  Options: --width=<32|64>, --host=<server>,
 * --table_ring[=slots] (both parties on one node: garbled tables stay in HBM, shared by hipIpc). */
#define _GNU_SOURCE
#include <errno.h>
#include <openssl/crypto.h>
#include <openssl/rand.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "baseot.h"
#include "protocol.h"

typedef struct { size_t n, next; const uint32_t *launch; double *time; double t0; } iter_marks;
static void note_launch(size_t i, void *ctx) {
    iter_marks *m = ctx;
    if (i == 0) host_trace_mark("first table evaluated");
    while (m->next < m->n && m->launch[m->next] == i) m->time[m->next++] = wall_clock() - m->t0;
}

int main(int argc, char **argv) {
    node *self = NULL;
    lgc_party *po = NULL;
    FILE *f = NULL;
    check(argc >= 7, "Usage: %s [Port] [Party] [Input file] [Algorithm] [Num. iterations CGD] [Precision]", argv[0]);
    char *algorithm = argv[4], *end;
    check(!strcmp(algorithm, "cholesky") || !strcmp(algorithm, "ldlt") || !strcmp(algorithm, "cgd"),
          "Algorithm must be cholesky, ldlt, or cgd.");
    errno = 0;
    int precision = (int)strtol(argv[6], &end, 10);
    check(!errno && !*end, "Precision must be a number");
    int party = !strcmp(argv[2], "1") ? 1 : (!strcmp(argv[2], "2") ? 2 : 0);
    check(party > 0, "Party must be either 1 or 2.");
    int w = 64, ring_slots = 0;
    long chunk_mb = 0;                       /* --table_chunk_mb=N: the largest launch's tables, in MiB (0: the default below) */
    const char *host = "localhost";
    for (int i = 7; i < argc; i++) {
        if (sscanf(argv[i], "--width=%i", &w) == 1) continue;
        if (!strcmp(argv[i], "--table_ring")) { ring_slots = TABLE_RING_BYTES; continue; }
        if (sscanf(argv[i], "--table_ring=%i", &ring_slots) == 1) continue;
        if (sscanf(argv[i], "--table_chunk_mb=%ld", &chunk_mb) == 1 && chunk_mb > 0) continue;
        { int lanes = 0; if (sscanf(argv[i], "--table_lanes=%i", &lanes) == 1) { protocol_set_table_lanes(lanes); continue; } }
        if (!strncmp(argv[i], "--host=", 7)) host = argv[i] + 7;
    }
    int num_iterations = !strcmp(algorithm, "cgd") ? atoi(argv[5]) : 0;
    int device = getenv("LINREG_DEVICE") ? atoi(getenv("LINREG_DEVICE")) : 0;

    /* read_ls_from_file (:11-79): A as d x d, b as d, no normalisation */
    f = fopen(argv[3], "r");
    check(f, "Could not open file: %s.", strerror(errno));
    size_t d, d2, dl;
    check(fscanf(f, "%zu %zu", &d, &d2) == 2 && d == d2, "Could not read A.");
    int64_t *Af = malloc(d * d * 8), *bf = malloc(d * 8);
    check(!read_values(f, d * d, precision, 1.0, w, Af), "Could not read A.");
    check(fscanf(f, "%zu", &dl) == 1 && dl == d, "Could not read b.");
    check(!read_values(f, d, precision, 1.0, w, bf), "Could not read b.");
    fclose(f); f = NULL;
    const size_t T = d * (d + 1) / 2;
    const uint64_t mw = w == 32 ? 0xffffffffull : ~0ull;
    uint64_t *mine = malloc((T + d) * 8);
    for (size_t i = 0; i < d; i++) {
        for (size_t j = 0; j <= i; j++) {
            uint64_t mask = 123456;
            mine[idx(i, j)] = (party == 1 ? mask : (uint64_t)Af[i * d + j] - mask) & mw;
        }
        uint64_t bm = 0xDEADBEEFull;
        mine[T + i] = (party == 1 ? bm : (uint64_t)bf[i] - bm) & mw;
    }

    char ep1[300], ep2[300], *eps[2] = {ep1, ep2};
    snprintf(ep1, sizeof ep1, "%s:%s", host, argv[1]);
    snprintf(ep2, sizeof ep2, "%s:0", host);
    check(!node_new(&self, party, 2, eps), "Could not connect");
    double time = wall_clock();
    if (party == 2) printf("\nAlgorithm: %s\n", algorithm);
    /* "check if inputs have equal dimensions" (src/linear.oc:109-114): revealOblivBool(feedOblivInt(d, 1) == feedOblivInt(d, 2)).
     * As in the reference the two d are compared INSIDE a circuit -- a program of its own (LGC_ALG_DIMCHECK: one comparison of
     * two 32-bit words, 31 AND gates, one reveal), because the program of the solve can only be built once both sides agree on
     * d -- and the bit is revealed to both.  The base OTs come first: the evaluator's d enters through the same OT extension
     * session as its shares do afterwards.  (Rounds 1-3 exchanged the two d on the socket.) */
    uint8_t delta[16], seeds[128][16], s0[128][16], s1[128][16];
    lgc_ot_sender *S = 0;
    lgc_ot_receiver *R = 0;
    unsigned long long dim_gates = 0;
    if (party == 1) { check(!baseot_ext_sender(self, 2, delta, seeds), "base OT failed"); LGC(lgc_ot_sender_create(&S, device, delta, seeds)); }
    else { check(!baseot_ext_receiver(self, 1, s0, s1), "base OT failed"); LGC(lgc_ot_receiver_create(&R, device, s0, s1)); }
    {
        lgc_system ds;
        memset(&ds, 0, sizeof ds);
        ds.d = 1; ds.width = 32; ds.precision = 0; ds.algorithm = LGC_ALG_DIMCHECK; ds.nshares = 2;
        lgc_party *pc = 0;
        uint64_t din[2] = {(uint64_t)(uint32_t)d, 0};
        uint8_t equal = 0;
        if (party == 1) {
            uint8_t sd[16];
            check(RAND_bytes(sd, sizeof sd) == 1, "RAND_bytes failed");
            LGC(lgc_party_create(&pc, device, &ds, LGC_ROLE_GARBLER, sd, (size_t)64 << 20));
            size_t bits = lgc_party_input_bits(pc);
            uint8_t *lab = malloc(bits * 16), *m0 = malloc(bits * 16), *m1 = malloc(bits * 16), *u = malloc(lgc_ot_u_bytes(bits)), *e = malloc(bits * 32);
            LGC(lgc_party_encode_inputs(pc, 0, din, lab));
            check(!send_blob(self, 2, lab, bits * 16), "could not send labels");
            LGC(lgc_party_input_pairs(pc, 1, m0, m1));
            check(!recv_blob(self, 2, u, lgc_ot_u_bytes(bits)), "OT: could not receive u");
            LGC(lgc_ot_labels_send(S, m0, m1, bits, u, e));
            check(!send_blob(self, 2, e, bits * 32), "OT: could not send");
            free(lab); free(m0); free(m1); free(u); free(e);
            for (size_t i = 0; i < lgc_party_num_launches(pc); i++) {
                size_t tb = lgc_party_table_bytes(pc, i);
                uint8_t *tab = malloc(tb + 16);
                LGC(lgc_party_garble(pc, i, tab));
                check(!send_blob(self, 2, tab, tb), "could not send tables");
                free(tab);
            }
            uint64_t dec[4] = {0};
            LGC(lgc_party_decode_bits(pc, dec));
            check(!send_blob(self, 2, dec, lgc_party_num_reveal(pc) * 8), "could not send decode bits");
            check(!recv_blob(self, 2, &equal, 1), "could not receive the comparison");        /* revealed to both (party 0) */
        } else {
            LGC(lgc_party_create(&pc, device, &ds, LGC_ROLE_EVALUATOR, NULL, (size_t)64 << 20));
            size_t bits = lgc_party_input_bits(pc);
            uint8_t *lab = malloc(bits * 16), *sel = malloc(bits), *u = malloc(lgc_ot_u_bytes(bits)), *e = malloc(bits * 32);
            check(!recv_blob(self, 1, lab, bits * 16), "could not receive labels");
            LGC(lgc_party_set_input_labels(pc, 0, lab));
            for (size_t i = 0; i < 2; i++)
                for (int j = 0; j < 32; j++) sel[i * 32 + (size_t)j] = (uint8_t)((din[i] >> j) & 1);
            LGC(lgc_ot_labels_recv_start(R, sel, bits, u));
            check(!send_blob(self, 1, u, lgc_ot_u_bytes(bits)), "OT: could not send u");
            check(!recv_blob(self, 1, e, bits * 32), "OT: could not receive");
            LGC(lgc_ot_labels_recv_finish(R, e, lab));
            LGC(lgc_party_set_input_labels(pc, 1, lab));
            free(lab); free(sel); free(u); free(e);
            for (size_t i = 0; i < lgc_party_num_launches(pc); i++) {
                size_t tb = lgc_party_table_bytes(pc, i);
                uint8_t *tab = malloc(tb + 16);
                check(!recv_blob(self, 1, tab, tb), "could not receive tables");
                LGC(lgc_party_evaluate(pc, i, tab));
                free(tab);
            }
            uint64_t dec[4] = {0};
            int64_t eqw = 0;
            check(!recv_blob(self, 1, dec, lgc_party_num_reveal(pc) * 8), "could not receive decode bits");
            LGC(lgc_party_finish(pc, dec, &eqw, NULL, NULL));
            equal = (uint8_t)(eqw & 1);
            check(!send_blob(self, 1, &equal, 1), "could not send the comparison");
        }
        dim_gates = (unsigned long long)lgc_party_and_gates(pc);
        lgc_party_destroy(pc);
        check(equal == 1, "Inputs of the two parties differ.");
    }

    lgc_system sys;
    memset(&sys, 0, sizeof sys);
    sys.d = d; sys.width = w; sys.precision = precision;
    sys.algorithm = !strcmp(algorithm, "cholesky") ? LGC_ALG_CHOLESKY : (!strcmp(algorithm, "ldlt") ? LGC_ALG_LDLT : LGC_ALG_CGD);
    sys.num_iterations = num_iterations; sys.nshares = 2; sys.normalize = 0; sys.trace = 1;
    const size_t chunk = chunk_mb > 0 ? (size_t)chunk_mb << 20 : ring_slots > 0 ? (size_t)64 << 30 : (size_t)64 << 20;   /* (host/linreg.c: kTableChunk) */
    if (party == 1) {
        uint8_t seed[16];
        check(RAND_bytes(seed, sizeof seed) == 1, "RAND_bytes failed");
        LGC(lgc_party_create(&po, device, &sys, LGC_ROLE_GARBLER, seed, chunk));
        check(!tables_ring_prepare(po, ring_slots), "could not create the table ring");   /* now: not on the evaluator's clock */
        size_t bits = lgc_party_input_bits(po);
        uint8_t *lab = malloc(bits * 16), *m0 = malloc(bits * 16), *m1 = malloc(bits * 16), *u = malloc(lgc_ot_u_bytes(bits)), *e = malloc(bits * 32);
        LGC(lgc_party_encode_inputs(po, 0, mine, lab));                         /* feedOblivLLong(.., 1) */
        check(!send_blob(self, 2, lab, bits * 16), "could not send labels");
        LGC(lgc_party_input_pairs(po, 1, m0, m1));                              /* feedOblivLLong(.., 2): OT */
        check(!recv_blob(self, 2, u, lgc_ot_u_bytes(bits)), "OT: could not receive u");
        LGC(lgc_ot_labels_send(S, m0, m1, bits, u, e));
        check(!send_blob(self, 2, e, bits * 32), "OT: could not send");
        host_trace_mark("input OT done");
        check(!programs_agree(self, 2, po, 1), "program check failed");
        check(!tables_send(self, 2, po, ring_slots, chunk), "could not stream the garbled tables");
        /* the OT session (device and page-locked buffers) and the label pairs go AFTER the tables: released before them, as in
         * rounds 1-4, their clean-up -- hundreds of MB at d = 500 -- ran on the Evaluator's iteration clock (cgd.oc:190-194)
         * while the Evaluator waited for the first table */
        lgc_ot_sender_destroy(S);
        OPENSSL_cleanse(m0, bits * 16); OPENSSL_cleanse(m1, bits * 16);     /* both labels of every input bit: their XOR is R */
        free(lab); free(m0); free(m1); free(u); free(e);
        size_t nr = lgc_party_num_reveal(po);
        uint64_t *dec = malloc((nr + 1) * 8);
        LGC(lgc_party_decode_bits(po, dec));
        check(!send_blob(self, 2, dec, nr * 8), "could not send decode bits");
        free(dec);
    } else {
        double t0 = wall_clock();
        LGC(lgc_party_create(&po, device, &sys, LGC_ROLE_EVALUATOR, NULL, chunk));
        size_t bits = lgc_party_input_bits(po);
        uint8_t *lab = malloc(bits * 16), *sel = malloc(bits), *u = malloc(lgc_ot_u_bytes(bits)), *e = malloc(bits * 32);
        check(!recv_blob(self, 1, lab, bits * 16), "could not receive labels");
        LGC(lgc_party_set_input_labels(po, 0, lab));
        for (size_t i = 0; i < T + d; i++)
            for (int j = 0; j < w; j++) sel[i * (size_t)w + (size_t)j] = (uint8_t)((mine[i] >> j) & 1);
        LGC(lgc_ot_labels_recv_start(R, sel, bits, u));
        check(!send_blob(self, 1, u, lgc_ot_u_bytes(bits)), "OT: could not send u");
        check(!recv_blob(self, 1, e, bits * 32), "OT: could not receive");
        LGC(lgc_ot_labels_recv_finish(R, e, lab));
        LGC(lgc_party_set_input_labels(po, 1, lab));
        lgc_ot_receiver_destroy(R);
        free(lab); free(sel); free(u); free(e);
        double t_ot = wall_clock() - t0;
        const int is_cgd = sys.algorithm == LGC_ALG_CGD;
        size_t n_marks = is_cgd ? (size_t)num_iterations : 0;
        uint32_t *mark_launch = malloc((n_marks + 1) * sizeof *mark_launch);
        uint64_t *mark_gates = malloc((n_marks + 1) * sizeof *mark_gates);
        double *mark_time = malloc((n_marks + 1) * sizeof *mark_time);
        if (is_cgd) LGC(lgc_party_iteration_marks(po, mark_launch, mark_gates, n_marks));
        double t_iters = wall_clock();                                          /* cgd.oc: time_start */
        iter_marks marks = {n_marks, 0, mark_launch, mark_time, t_iters};
        check(!programs_agree(self, 1, po, 0), "program check failed");
        check(!tables_recv(self, 1, po, ring_slots, chunk, note_launch, &marks), "could not receive garbled tables");
        host_trace_mark("tables evaluated");
        size_t nr = lgc_party_num_reveal(po);
        uint64_t *dec = malloc((nr + 1) * 8);
        check(!recv_blob(self, 1, dec, nr * 8), "could not receive decode bits");
        int64_t *beta = malloc(d * 8), *trace = malloc(((size_t)num_iterations * (d + 4) + 1) * 8);
        LGC(lgc_party_finish(po, dec, beta, trace, NULL));
        free(dec);
        long long gates = (long long)lgc_party_and_gates(po) + (long long)dim_gates;   /* yaoGateCount() includes the dimension check */
        if (sys.algorithm == LGC_ALG_CGD) {
            printf("OT time: %f\nStarting iterations.\n", t_ot);
            for (int t = 0; t < num_iterations; t++) {
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
        printf("Time elapsed: %f\n", wall_clock() - time);
        printf("Number of gates: %lld\n", gates);
        {   /* the same solve in the reference's own circuit: keeps result files comparable with
             * experiments/results/phase2_{32,64}; nothing is printed where the reference published no count (ldlt) */
            uint64_t refg = 0;
            if (lgc_reference_gate_count(sys.algorithm, sys.width, d, sys.num_iterations, &refg) == 0)
                printf("Reference-equivalent gates: %llu\n", (unsigned long long)refg);
        }
        printf("Result: ");
        for (size_t i = 0; i < d; i++) printf("%20.15f ", fixed_to_double(beta[i], precision));
        printf("\n");
        free(beta); free(trace); free(mark_launch); free(mark_gates); free(mark_time);
    }
    lgc_party_destroy(po);
    node_destroy(&self);
    free(Af); free(bf); free(mine);
    return 0;
error:
    if (f) fclose(f);
    if (po) lgc_party_destroy(po);
    node_destroy(&self);
    return 1;
}
