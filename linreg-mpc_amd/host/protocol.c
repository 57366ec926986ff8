/* protocol.c -- what the host-side protocol files share: fixed-point conversion (src/fixed.c), the input file's numbers
 * (src/linear.c:27-102), the length-prefixed proto2 messages of phase 1 (src/phase1.c:100-145) and plain blobs.  The drivers
 * themselves: tables.c (phase 2 table stream), phase1_ti.c (trusted initializer, --ti_ring), phase1_party.c (data provider:
 * run_party) -- one 1 650-line file until round 4.  Every compute step runs on the MI355X through liblinreg_gc; these files
 * move bytes between parties. */
#define _GNU_SOURCE
#include <errno.h>
#include <malloc.h>
#include <math.h>
#include <openssl/rand.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"
#include "baseot.h"
#include "config.h"
#include "net.h"
#include "pmsg.h"
#include "protocol.h"


size_t idx(size_t i, size_t j) { if (j > i) { size_t t = i; i = j; j = t; } return (i * (i + 1)) / 2 + j; }

/* double_to_fixed / fixed_to_double / read_values / read_own_columns: readdata.c */

/* length-prefixed protobuf message (src/phase1.c:100-145) */
int send_pmsg(node *self, int to, const uint64_t *vec, size_t n, uint64_t value) {
    size_t sz = pmsg_packed_size(vec, n, value);
    uint8_t *buf = malloc(sz + sizeof(size_t));
    memcpy(buf, &sz, sizeof sz);
    pmsg_pack(vec, n, value, buf + sizeof sz);
    int rc = net_send_flush(self, to, buf, sz + sizeof sz);   /* flush: orecv(pd,0,NULL,0) at src/phase1.c:141 */
    free(buf);
    return rc;
}
/* the framed bytes of send_pmsg (8-byte length + protobuf), malloc'd */
uint8_t *frame_pmsg(const uint64_t *vec, size_t n, uint64_t value, size_t *len) {
    size_t sz = pmsg_packed_size(vec, n, value);
    uint8_t *buf = malloc(sz + sizeof(size_t));
    if (!buf) return NULL;
    memcpy(buf, &sz, sizeof sz);
    pmsg_pack(vec, n, value, buf + sizeof sz);
    *len = sz + sizeof sz;
    return buf;
}
/* A phase-1 message holds at most n varints of <= 10 bytes plus two tags, one length and `value`
 * (src/protobuf, the .proto files): a longer length prefix from a peer is rejected before anything is allocated. */
size_t g_pmsg_limit = (size_t)1 << 32;       /* bound on the length prefix recv_pmsg accepts (pmsg_set_limit) */
void pmsg_set_limit(size_t n_elements) { g_pmsg_limit = 10 * n_elements + 64; }
int recv_pmsg(node *self, int from, uint64_t **vec, size_t *n, uint64_t *value) {
    size_t sz = 0;
    if (net_recv(self, from, &sz, sizeof sz)) return 1;
    if (sz > g_pmsg_limit) { fprintf(stderr, "message of %zu bytes from party %d exceeds the protocol's bound\n", sz, from); return 1; }
    uint8_t *buf = malloc(sz ? sz : 1);
    if (!buf) return 1;
    if (net_recv(self, from, buf, sz)) { free(buf); return 1; }
    int rc = pmsg_unpack(buf, sz, vec, n, value);
    free(buf);
    return rc;
}
int send_blob(node *self, int to, const void *buf, uint64_t len) {
    if (net_send(self, to, &len, sizeof len)) return 1;
    return len ? net_send(self, to, buf, len) : 0;
}
int recv_blob(node *self, int from, void *buf, uint64_t len) {
    uint64_t got = 0;
    if (net_recv(self, from, &got, sizeof got) || got != len) return 1;
    return len ? net_recv(self, from, buf, len) : 0;
}
