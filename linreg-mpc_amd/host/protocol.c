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
#include "baseot.h"
#include "config.h"
#include "net.h"
#include "pmsg.h"
#include "protocol.h"


size_t idx(size_t i, size_t j) { if (j > i) { size_t t = i; i = j; j = t; } return (i * (i + 1)) / 2 + j; }

/* (fixed_t)(d * (1ll << p)) with the phase-2 type (src/fixed.c:3-5, src/linear.c:51) */
/* Out of range the C cast is undefined; the reference runs on x86-64, where cvttsd2si yields the
 * "integer indefinite" value (INT_MIN of the type) -- the rule the oracle states (orc_double_to_fixed). */
int64_t double_to_fixed(double d, int p, int w) {
    double t = d * (double)(1ll << p);
    if (w == 32) {
        if (!(t > -2147483649.0 && t < 2147483648.0)) return (int64_t)INT32_MIN;
        return (int64_t)(int32_t)t;
    }
    if (!(t >= -9223372036854775808.0 && t < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)t;
}
double fixed_to_double(int64_t f, int p) { return ((double)f) / (double)(1ll << p); }

/* read_matrix / read_vector (src/linear.c:27-102), values divided by the normalizer */
int read_values(FILE *f, size_t count, int precision, double normalizer, int w2, int64_t *out) {
    for (size_t i = 0; i < count; i++) {
        double val;
        if (fscanf(f, "%lf", &val) != 1) return 1;
        val /= normalizer;
        out[i] = double_to_fixed(val, precision, w2);
    }
    return 0;
}

/* read_matrix / read_vector for a data provider: the rest of the input file is slurped and scanned token by
 * token; only the columns this party owns (and the target, if it owns it) are converted -- with strtod, i.e. the
 * same correctly rounded value "%lf" gives -- and quantised, every other entry stays 0 (it is never used: a
 * provider only ever touches its own columns).  Token counts and syntax are still checked for the whole file. */
static int is_num_char(int ch) { return (ch >= '0' && ch <= '9') || ch == '.' || ch == '-' || ch == '+' || ch == 'e' || ch == 'E' || ch == 'i' || ch == 'n' || ch == 'f' || ch == 'a' || ch == 'I' || ch == 'N' || ch == 'F' || ch == 'A' || ch == 'x' || ch == 'X'; }
int read_own_columns(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
                            int64_t *Xq, int64_t *yq) {
    long at = ftell(f);
    if (at < 0 || fseek(f, 0, SEEK_END)) return 1;
    long end = ftell(f);
    if (end < at || fseek(f, at, SEEK_SET)) return 1;
    size_t len = (size_t)(end - at);
    char *buf = malloc(len + 1);
    if (!buf || fread(buf, 1, len, f) != len) { free(buf); return 1; }
    buf[len] = 0;
    char *p = buf;
    int rc = 1;
#define SKIP_WS() while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f') p++
    size_t n2 = 0, d2 = 0;
    { char *e; SKIP_WS(); n2 = strtoull(p, &e, 10); if (e == p) goto out; p = e; SKIP_WS(); d2 = strtoull(p, &e, 10); if (e == p) goto out; p = e; }
    if (n2 != n || d2 != d) goto out;
    memset(Xq, 0, n * d * sizeof *Xq);
    for (size_t k = 0; k < n; k++)
        for (size_t j = 0; j < d; j++) {
            SKIP_WS();
            if (!*p) goto out;
            if (j >= c0 && j < c1) {
                char *e;
                double v = strtod(p, &e);
                if (e == p) goto out;
                p = e;
                Xq[k * d + j] = double_to_fixed(v / normalizer, precision, w2);
            } else {
                if (!is_num_char((unsigned char)*p)) goto out;       /* first character checked, the rest of the token skipped */
                while ((unsigned char)*p > ' ') p++;
            }
        }
    { char *e; SKIP_WS(); n2 = strtoull(p, &e, 10); if (e == p || n2 != n) goto out; p = e; }
    memset(yq, 0, n * sizeof *yq);
    for (size_t k = 0; k < n; k++) {
        SKIP_WS();
        if (!*p) goto out;
        char *e;
        double v = strtod(p, &e);
        if (e == p) goto out;
        p = e;
        if (own_y) yq[k] = double_to_fixed(v / normalizer, precision, w2);
    }
#undef SKIP_WS
    rc = 0;
out:
    free(buf);
    return rc;
}

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
