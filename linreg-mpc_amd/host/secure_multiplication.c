/* secure_multiplication.c -- phase-1-only benchmark driver with the reference's command line and
 * JSON timing lines (src/cmd/secure_multiplication.c:40-116):
 *     secure_multiplication file precision party [--use_ot] [--width_phase1=<32|64>]
 * party 1 prints {"n":..,"d":..,"p":..}; every party prints
 *     {"party":"k", "cputime":"..", "wait_time":.., "realtime":".."}
 * plus one extra line with the bytes sent per peer (what Obliv-C's -DPROFILE_NETWORK build reports,
 * experiments/test_phase1_aws.py:245-252). */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "protocol.h"

int main(int argc, char **argv) {
    config *c = NULL;
    node *self = NULL;
    uint64_t *sa = NULL, *sb = NULL;
    struct timespec c0, c1, r0, r1;
    check(argc > 3, "Usage: %s file precision party [options]\nOptions: --use_ot: Enables the OT-based phase 1 protocol", argv[0]);
    char *end;
    errno = 0;
    int precision = (int)strtol(argv[2], &end, 10);
    check(!errno && !*end, "Precision must be a number");
    int party = (int)strtol(argv[3], &end, 10);
    check(!errno && !*end, "Party must be a number");
    int use_ot = 0, w1 = 64;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--use_ot")) use_ot = 1;
        else if (sscanf(argv[i], "--width_phase1=%i", &w1) == 1) {}
    }
    int device = getenv("LINREG_DEVICE") ? atoi(getenv("LINREG_DEVICE")) : 0;
    check(!config_new(&c, argv[1]), "Could not read config");
    c->party = party;
    if (party == 1) printf("{\"n\":\"%zd\", \"d\":\"%zd\", \"p\":\"%d\"}\n", c->n, c->d, c->num_parties - 2);
    check(!node_new(&self, party, c->num_parties, c->endpoint), "Could not create node");
    check(!net_barrier(self), "barrier failed");                 /* wait until everybody has started up */
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c0);
    clock_gettime(CLOCK_MONOTONIC, &r0);
    if (party == 1) {
        if (!use_ot) check(!run_trusted_initializer(self, c, w1, device), "Error while running trusted initializer");
    } else if (party > 2) {
        check(!run_party(self, c, precision, precision, w1, w1, use_ot, device, &sa, &sb), "Error while running party %d", party);
    }
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c1);
    clock_gettime(CLOCK_MONOTONIC, &r1);
    double bill = 1e9, wait_s = 0.0;      /* summed over the per-peer worker threads */
    for (int q = 0; q < self->num_parties; q++) wait_s += (double)self->wait_ns[q] / bill;
    printf("{\"party\":\"%d\", \"cputime\":\"%f\", \"wait_time\":%f, \"realtime\":\"%f\"}\n", party,
           (c1.tv_sec - c0.tv_sec) + (c1.tv_nsec - c0.tv_nsec) / bill, wait_s,
           (r1.tv_sec - r0.tv_sec) + (r1.tv_nsec - r0.tv_nsec) / bill);
    printf("{\"party\":\"%d\", \"bytes_sent\":[", party);
    for (int q = 0; q < self->num_parties; q++) printf("%s%llu", q ? ", " : "", (unsigned long long)self->sent[q]);
    printf("], \"sends\":[");
    for (int q = 0; q < self->num_parties; q++) printf("%s%llu", q ? ", " : "", (unsigned long long)self->nsend[q]);
    printf("]}\n");
    check(!net_barrier(self), "barrier failed");
    /* what Obliv-C built with -DPROFILE_NETWORK prints per connection at cleanup, in party order -- the lines
     * experiments/test_phase1_aws.py:245-250 collects into the [bytes...] / [flushes...] rows of the .out files */
    for (int q = 1; q <= self->num_parties; q++) {
        if (q == party) continue;
        printf("Total bytes sent: %llu\n", (unsigned long long)self->sent[q - 1]);
        printf("Total flush done: %llu\n", (unsigned long long)net_flush_count(self, q));
    }
    node_destroy(&self);
    config_destroy(&c);
    free(sa); free(sb);
    return 0;
error:
    config_destroy(&c);
    node_destroy(&self);
    return 1;
}
