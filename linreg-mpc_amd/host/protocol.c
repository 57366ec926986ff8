/* protocol.c -- the phase-1 drivers shared by bin/linreg and bin/secure_multiplication
 * (counterparts of run_trusted_initializer / run_party, src/phase1.c:241-656): every compute
 * step runs on the MI355X through liblinreg_gc; this file moves bytes between parties. */
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
static int read_own_columns(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
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
static uint8_t *frame_pmsg(const uint64_t *vec, size_t n, uint64_t value, size_t *len) {
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
static size_t g_pmsg_limit = (size_t)1 << 32;
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

/* ---------------------------------------------------------------- phase 2: table stream */
#define TCHK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, lgc_last_error()); return 1; } } while (0)
/* a trace mark of the host (LINREG_TRACE); in bin/linreg_testhooks also the place where LINREG_DIE_AT=<mark> makes this
 * party kill itself (tests/test_host.py: a party lost at a known point of the protocol) */
void host_trace_mark(const char *what) {
    lgc_trace_mark(what);
#ifdef LINREG_TEST_HOOKS
    const char *die = getenv("LINREG_DIE_AT");
    if (die && !strcmp(die, what)) raise(SIGKILL);
#endif
}

typedef struct { uint8_t handle[64]; uint64_t nslots, slot_bytes; } ring_hello;

/* Two-stage pipeline between the GPU and the socket of the table stream: kTableSlots page-locked buffers of
 * one launch each; `head` launches have been filled (garbled / received), `tail` have been drained (sent /
 * evaluated).  The reference's Yao protocol overlaps nothing here (osend per gate, bcipher/yao), and a serial
 * garble -> copy -> send -> next loop leaves the GPU, the PCIe link and the socket each idle two thirds of the time. */
enum { kTableSlots = 3, kMaxLanes = 16 };
/* --table_lanes=K: the table bytes of a launch are striped over K extra TCP connections (one stream moves what one core
 * copies on either side: 7.9 GB/s on loopback); 0 = the party connection alone, length-prefixed as every other message */
static int g_table_lanes = 0;
void protocol_set_table_lanes(int k) { g_table_lanes = k < 0 ? 0 : (k > kMaxLanes ? kMaxLanes : k); }
typedef struct {
    node *self; int peer; lgc_party *po; size_t nl;
    uint8_t *buf[kTableSlots]; int pinned;
    size_t head, tail; int failed;
    int lanes, fd[kMaxLanes];          /* lanes = 0: the party connection */
    int cnt[kTableSlots];              /* workers through with the launch in this slot */
    pthread_mutex_t mu; pthread_cond_t cv;
} table_pipe;
typedef struct { table_pipe *t; int lane; } table_worker;
static void table_pipe_free(table_pipe *t) {
    for (int k = 0; k < kTableSlots; k++) { if (t->pinned) lgc_host_free(t->buf[k]); else free(t->buf[k]); t->buf[k] = NULL; }
    for (int l = 0; l < t->lanes; l++) if (t->fd[l] >= 0) close(t->fd[l]);
    pthread_mutex_destroy(&t->mu); pthread_cond_destroy(&t->cv);
}
static int table_pipe_init(table_pipe *t, node *self, int peer, lgc_party *po, size_t nl, size_t chunk) {
    memset(t, 0, sizeof *t);
    t->self = self; t->peer = peer; t->po = po; t->nl = nl;
    size_t biggest = 0;
    for (size_t i = 0; i < nl; i++) { size_t b = lgc_party_table_bytes(po, i); if (b > biggest) biggest = b; }
    if (biggest < chunk) biggest = chunk;
    pthread_mutex_init(&t->mu, NULL); pthread_cond_init(&t->cv, NULL);
    t->pinned = 1;
    for (int k = 0; k < kTableSlots; k++) {
        t->buf[k] = lgc_host_alloc(biggest + 4096);
        if (!t->buf[k]) { fprintf(stderr, "table stream: %s\n", lgc_last_error()); table_pipe_free(t); return 1; }
    }
    return 0;
}
static void table_pipe_fail(table_pipe *t) {
    pthread_mutex_lock(&t->mu); t->failed = 1; pthread_cond_broadcast(&t->cv); pthread_mutex_unlock(&t->mu);
}
static int table_pipe_failed(table_pipe *t) {
    pthread_mutex_lock(&t->mu); int f = t->failed; pthread_mutex_unlock(&t->mu); return f;
}
/* producer side of buffer i % kTableSlots: wait until launch i - kTableSlots has been drained */
static uint8_t *table_pipe_acquire(table_pipe *t, size_t i) {
    pthread_mutex_lock(&t->mu);
    while (!t->failed && i >= t->tail + kTableSlots) pthread_cond_wait(&t->cv, &t->mu);
    int f = t->failed;
    pthread_mutex_unlock(&t->mu);
    return f ? NULL : t->buf[i % kTableSlots];
}
static void table_pipe_publish(table_pipe *t) {
    pthread_mutex_lock(&t->mu); t->head++; pthread_cond_broadcast(&t->cv); pthread_mutex_unlock(&t->mu);
}
/* consumer side: wait until launch i has been filled */
static uint8_t *table_pipe_take(table_pipe *t, size_t i) {
    pthread_mutex_lock(&t->mu);
    while (!t->failed && t->head <= i) pthread_cond_wait(&t->cv, &t->mu);
    int f = t->failed;
    pthread_mutex_unlock(&t->mu);
    return f ? NULL : t->buf[i % kTableSlots];
}
static void table_pipe_release(table_pipe *t) {
    pthread_mutex_lock(&t->mu); t->tail++; pthread_cond_broadcast(&t->cv); pthread_mutex_unlock(&t->mu);
}
/* one of the socket workers is through with launch i: the last one hands the slot on.  The workers take the launches in
 * order and a slot is only refilled once it has been handed on, so one counter per slot is enough. */
static void table_pipe_worker_done(table_pipe *t, size_t i, int sending) {
    const int workers = t->lanes > 0 ? t->lanes : 1;
    pthread_mutex_lock(&t->mu);
    if (++t->cnt[i % kTableSlots] == workers) {
        t->cnt[i % kTableSlots] = 0;
        if (sending) t->tail++; else t->head++;
        pthread_cond_broadcast(&t->cv);
    }
    pthread_mutex_unlock(&t->mu);
}
/* the stripe of a launch that lane l carries (4 KiB granules) */
static void lane_stripe(size_t len, int lanes, int l, size_t *off, size_t *n) {
    size_t seg = (((len + (size_t)lanes - 1) / (size_t)lanes) + 4095) & ~(size_t)4095;
    size_t o = (size_t)l * seg;
    if (o > len) o = len;
    *off = o;
    *n = len - o < seg ? len - o : seg;
}
static void *table_pipe_sender(void *arg) {
    table_worker *w = arg;
    table_pipe *t = w->t;
    for (size_t i = 0; i < t->nl; i++) {
        uint8_t *tab = table_pipe_take(t, i);
        if (!tab) break;
        const size_t len = lgc_party_table_bytes(t->po, i);
        int bad;
        if (t->lanes == 0) {
            bad = send_blob(t->self, t->peer, tab, len);
        } else {
            size_t off, n;
            lane_stripe(len, t->lanes, w->lane, &off, &n);
            bad = n ? net_io_all(t->fd[w->lane], tab + off, n, 1) : 0;
            if (!bad) __atomic_fetch_add(&t->self->sent[t->peer - 1], n, __ATOMIC_RELAXED);
        }
        if (bad) { table_pipe_fail(t); break; }
        table_pipe_worker_done(t, i, 1);
    }
    return NULL;
}
static void *table_pipe_receiver(void *arg) {
    table_worker *w = arg;
    table_pipe *t = w->t;
    for (size_t i = 0; i < t->nl; i++) {
        uint8_t *tab = table_pipe_acquire(t, i);
        if (!tab) break;
        const size_t len = lgc_party_table_bytes(t->po, i);
        int bad;
        if (t->lanes == 0) {
            bad = recv_blob(t->self, t->peer, tab, len);
        } else {
            size_t off, n;
            lane_stripe(len, t->lanes, w->lane, &off, &n);
            bad = n ? net_io_all(t->fd[w->lane], tab + off, n, 0) : 0;
        }
        if (bad) { table_pipe_fail(t); break; }
        table_pipe_worker_done(t, i, 0);
    }
    return NULL;
}
/* start / stop the socket workers of a pipe */
static int table_pipe_start(table_pipe *t, void *(*fn)(void *), pthread_t *th, table_worker *w) {
    const int workers = t->lanes > 0 ? t->lanes : 1;
    for (int l = 0; l < workers; l++) {
        w[l].t = t; w[l].lane = l;
        if (pthread_create(&th[l], NULL, fn, &w[l])) {
            table_pipe_fail(t);                           /* the workers already running see `failed` and leave */
            for (int j = 0; j < l; j++) pthread_join(th[j], NULL);
            return 1;
        }
    }
    return 0;
}
static void table_pipe_stop(table_pipe *t, pthread_t *th) {
    const int workers = t->lanes > 0 ? t->lanes : 1;
    /* No pthread_cancel: a worker cancelled inside pthread_cond_wait would die holding t->mu and hang everyone else.
     * After a failure the condition variable has been broadcast (workers in table_pipe_take / _acquire see `failed` and
     * leave); a worker inside send() / recv() on a dead peer is released by shutting its socket down. */
    if (table_pipe_failed(t)) {
        for (int l = 0; l < t->lanes; l++) if (t->fd[l] >= 0) shutdown(t->fd[l], SHUT_RDWR);
        if (t->lanes == 0 && t->self && t->peer >= 1 && t->self->fd[t->peer - 1] >= 0) shutdown(t->self->fd[t->peer - 1], SHUT_RDWR);
    }
    for (int l = 0; l < workers; l++) pthread_join(th[l], NULL);
}

/* ---- ring mode as a link: one (garbler block, evaluator block) pair, its hipIpc ring, and the byte channel that carries
 * the 80-byte hello and the one-byte ready / ack tokens -- the party connection (fd < 0), or a connection of its own
 * when several blocks of a sweep run side by side on several GPUs (bin/linreg --devices: one link and one thread per
 * device).  A link handles the launches [start, end) of its party object, in any number of consecutive ranges. */
static int link_io(table_link *l, void *buf, size_t n, int wr) {
    if (l->fd < 0) return wr ? send_blob(l->self, l->peer, buf, n) : recv_blob(l->self, l->peer, buf, n);
    return net_io_all(l->fd, buf, n, wr);
}
/* The garbler may create its ring ahead of time (tables_ring_prepare: while the other parties are still in phase 1 or in
 * the label OT): a device allocation of tens of GB takes from 0.1 s to seconds, and taken inside tables_send it is on the
 * evaluator's clock.  A prepared ring is picked up by table_link_open. */
enum { kMaxPrepared = 16 };
static struct { lgc_party *po; ring_hello h; } g_prepared[kMaxPrepared];
static pthread_mutex_t g_prepared_mu = PTHREAD_MUTEX_INITIALIZER;
static int ring_create(lgc_party *po, int ring_slots, ring_hello *h) {
    size_t sb = 0;
    if (ring_slots == TABLE_RING_BYTES) {
        TCHK(lgc_party_ring_create_bytes(po, 0, h->handle, &sb));
        h->nslots = 0;
    } else {
        TCHK(lgc_party_ring_create(po, ring_slots, h->handle, &sb));
        h->nslots = (uint64_t)ring_slots;
    }
    h->slot_bytes = sb;
    return 0;
}
int tables_ring_prepare(lgc_party *po, int ring_slots) {
    if (ring_slots <= 0) return 0;
    ring_hello h;
    size_t sb = 0;
    memset(&h, 0, sizeof h);
    if (ring_create(po, ring_slots, &h)) return 1;
    (void)sb;
    pthread_mutex_lock(&g_prepared_mu);
    int ok = 0;
    for (int i = 0; i < kMaxPrepared && !ok; i++) if (!g_prepared[i].po) { g_prepared[i].po = po; g_prepared[i].h = h; ok = 1; }
    pthread_mutex_unlock(&g_prepared_mu);
    return ok ? 0 : 1;
}
static int take_prepared(lgc_party *po, ring_hello *h) {
    int found = 0;
    pthread_mutex_lock(&g_prepared_mu);
    for (int i = 0; i < kMaxPrepared && !found; i++) if (g_prepared[i].po == po) { *h = g_prepared[i].h; g_prepared[i].po = NULL; found = 1; }
    pthread_mutex_unlock(&g_prepared_mu);
    return found;
}
int table_link_open(table_link *l, node *self, int peer, int fd, lgc_party *po, int sending, int ring_slots, size_t start) {
    memset(l, 0, sizeof *l);
    l->self = self; l->peer = peer; l->fd = fd; l->po = po; l->start = start; l->end = lgc_party_num_launches(po);
    ring_hello h;
    memset(&h, 0, sizeof h);
    if (sending) {
        if (!take_prepared(po, &h) && ring_create(po, ring_slots, &h)) return 1;
        if (link_io(l, &h, sizeof h, 1)) return 1;
    } else {
        if (link_io(l, &h, sizeof h, 0)) return 1;
        if (h.nslots == 0) TCHK(lgc_party_ring_open_bytes(po, h.handle, (size_t)h.slot_bytes));
        else TCHK(lgc_party_ring_open(po, h.handle, (int)h.nslots, (size_t)h.slot_bytes));
    }
    l->nslots = (size_t)h.nslots;
    return 0;
}
/* byte ring: the newest launch any launch of this link waits for (-1: none) -- the same on both sides (same plan) */
static int64_t link_last_ack(table_link *l) {
    if (l->last_ack_known) return l->last_ack;
    int64_t m = -1;
    for (size_t i = l->start; i < l->end; i++) {
        int64_t wf = lgc_party_ring_wait_for(l->po, i);
        if (wf > m) m = wf;
    }
    l->last_ack = m; l->last_ack_known = 1;
    return m;
}
/* garbler: launches [lo, hi); launch i reuses the slot of launch i - nslots and waits for its ack */
int table_link_send_range(table_link *l, size_t lo, size_t hi) {
    uint8_t tok = 0;
    if (l->nslots == 0) {
        /* byte ring: launch i may overwrite its range once the launch lgc_party_ring_wait_for names has been evaluated
         * (launches before l->start never pass through this link).  The evaluator acknowledges the launches somebody will
         * wait for -- those up to link_last_ack -- and no others: the garbler is through when its last launch is garbled,
         * not when the evaluator is (its exit handlers then run beside the evaluator's tail instead of beside its exit) */
        const int64_t last_ack = link_last_ack(l);
        for (size_t i = lo; i < hi; i++) {
            int64_t wf = lgc_party_ring_wait_for(l->po, i);
            size_t need = wf >= (int64_t)l->start ? (size_t)(wf - (int64_t)l->start) + 1 : 0;
            while (l->acked < need) { if (link_io(l, &tok, 1, 0)) return 1; l->acked++; }
            TCHK(lgc_party_garble_ring(l->po, i));
            if (i == 0) host_trace_mark("first table garbled");
            tok = 1;
            if (link_io(l, &tok, 1, 1)) return 1;
        }
        /* acknowledgements of this range that are still on their way stay out of the next message on this channel */
        size_t due = (last_ack >= (int64_t)l->start) ? (size_t)(last_ack - (int64_t)l->start) + 1 : 0;
        if (due > hi - l->start) due = hi - l->start;
        while (l->acked < due) { if (link_io(l, &tok, 1, 0)) return 1; l->acked++; }
        return 0;
    }
    for (size_t i = lo; i < hi; i++) {
        if (i - l->start >= l->nslots && link_io(l, &tok, 1, 0)) return 1;      /* slot is free again */
        TCHK(lgc_party_garble_ring(l->po, i));
        if (i == 0) host_trace_mark("first table garbled");
        tok = 1;
        if (link_io(l, &tok, 1, 1)) return 1;
    }
    return 0;
}
int table_link_recv_range(table_link *l, size_t lo, size_t hi, void (*after_launch)(size_t launch, void *ctx), void *ctx) {
    uint8_t tok = 0;
    for (size_t i = lo; i < hi; i++) {
        if (link_io(l, &tok, 1, 0)) return 1;                                   /* launch i is in its slot */
        TCHK(lgc_party_evaluate_ring(l->po, i));
        if (after_launch) after_launch(i, ctx);
        if ((l->nslots == 0 ? (int64_t)i <= link_last_ack(l) : i + l->nslots < l->end) && link_io(l, &tok, 1, 1)) return 1;
    }
    return 0;
}

int programs_agree(node *self, int peer, lgc_party *po, int sending) {
    uint8_t mine[32], theirs[32], ok = 0;
    if (lgc_party_program_fingerprint(po, mine) != 0) { fprintf(stderr, "%s\n", lgc_last_error()); return 1; }
    if (sending) {
        if (send_blob(self, peer, mine, sizeof mine) || recv_blob(self, peer, &ok, 1)) return 1;
    } else {
        if (recv_blob(self, peer, theirs, sizeof theirs)) return 1;
        ok = memcmp(mine, theirs, sizeof mine) == 0;
        if (send_blob(self, peer, &ok, 1)) return 1;
    }
    if (!ok)
        fprintf(stderr, "the CSP and the Evaluator built different programs: algorithm, iterations, precision, widths, --lambdas, "
                        "--gate_hash and --devices must be the same on parties 1 and 2\n");
    return ok ? 0 : 1;
}

int tables_send(node *self, int peer, lgc_party *po, int ring_slots, size_t chunk) {
    const size_t nl = lgc_party_num_launches(po);
    if (ring_slots > 0) {
        table_link l;
        if (table_link_open(&l, self, peer, -1, po, 1, ring_slots, 0)) return 1;
        return table_link_send_range(&l, 0, nl);
    }
    /* socket mode: launch i + 1 is garbled and copied out while launch i is on the wire */
    table_pipe tp;
    if (table_pipe_init(&tp, self, peer, po, nl, chunk)) return 1;
    if (net_lanes_offer(self, peer, g_table_lanes, tp.fd)) { fprintf(stderr, "table stream: could not open %d lanes\n", g_table_lanes); table_pipe_free(&tp); return 1; }
    tp.lanes = g_table_lanes;
    pthread_t th[kMaxLanes];
    table_worker tw[kMaxLanes];
    if (table_pipe_start(&tp, table_pipe_sender, th, tw)) { table_pipe_free(&tp); return 1; }
    for (size_t i = 0; i < nl && !table_pipe_failed(&tp); i++) {
        uint8_t *tab = table_pipe_acquire(&tp, i);           /* waits until the workers are through with this slot */
        if (!tab) break;
        if (lgc_party_garble(po, i, tab) != 0) { fprintf(stderr, "%s\n", lgc_last_error()); table_pipe_fail(&tp); break; }
        if (i == 0) host_trace_mark("first table garbled");
        table_pipe_publish(&tp);
    }
    table_pipe_stop(&tp, th);
    int rc = table_pipe_failed(&tp);
    table_pipe_free(&tp);
    return rc;
}

int tables_recv(node *self, int peer, lgc_party *po, int ring_slots, size_t chunk,
                void (*after_launch)(size_t launch, void *ctx), void *ctx) {
    const size_t nl = lgc_party_num_launches(po);
    if (ring_slots > 0) {
        table_link l;
        if (table_link_open(&l, self, peer, -1, po, 0, ring_slots, 0)) return 1;
        return table_link_recv_range(&l, 0, nl, after_launch, ctx);
    }
    /* socket mode: launch i + 1 is read from the socket while launch i is copied in and evaluated */
    table_pipe tp;
    if (table_pipe_init(&tp, self, peer, po, nl, chunk)) return 1;
    if (net_lanes_accept_offer(self, peer, kMaxLanes, &tp.lanes, tp.fd)) { fprintf(stderr, "table stream: could not open the lanes\n"); tp.lanes = 0; table_pipe_free(&tp); return 1; }
    pthread_t th[kMaxLanes];
    table_worker tw[kMaxLanes];
    if (table_pipe_start(&tp, table_pipe_receiver, th, tw)) { table_pipe_free(&tp); return 1; }
    for (size_t i = 0; i < nl; i++) {
        const uint8_t *tab = table_pipe_take(&tp, i);        /* waits until launch i has arrived */
        if (!tab) break;
        if (lgc_party_evaluate(po, i, tab) != 0) { fprintf(stderr, "%s\n", lgc_last_error()); table_pipe_fail(&tp); break; }
        table_pipe_release(&tp);
        if (after_launch) after_launch(i, ctx);
    }
    table_pipe_stop(&tp, th);
    int rc = table_pipe_failed(&tp);
    table_pipe_free(&tp);
    return rc;
}

/* ---------------------------------------------------------------- phase 1: trusted initializer */
/* Per batch the TI first encodes every message on a pool of threads (varint packing is the CPU cost),
 * then one sender thread per data provider writes that provider's messages in loop order. */
typedef struct { uint8_t *buf; size_t len; } ti_frame;
typedef struct {
    size_t n, nb, first, stride;      /* this thread encodes messages first, first + stride, ... of 2 * nb */
    const uint64_t *x, *y, *r, *xyr;
    ti_frame *frames;                 /* [2q] = (y, <x,y> - r) for party a, [2q + 1] = (x, r) for party b */
    int failed;
} ti_encoder;
static void *ti_encoder_main(void *arg) {
    ti_encoder *t = arg;
    for (size_t m = t->first; m < 2 * t->nb; m += t->stride) {
        size_t q = m >> 1;
        ti_frame *f = &t->frames[m];
        f->buf = (m & 1) ? frame_pmsg(t->x + q * t->n, t->n, t->r[q], &f->len)
                         : frame_pmsg(t->y + q * t->n, t->n, t->xyr[q], &f->len);
        if (!f->buf) t->failed = 1;
    }
    return NULL;
}
/* A ring of encoded batches decouples the destinations: the main thread generates and encodes batch
 * after batch; every data provider has a persistent sender thread that walks the batches at the pace
 * of ITS socket (a provider whose queues are full must not stall the messages of the others); a slot
 * is reused once all senders are through with it. */
enum { kTiRing = 8 };
typedef struct {
    ti_frame *frames;          /* 2 * batch frames */
    size_t q0, nb;             /* pairs [q0, q0 + nb) */
    int done;                  /* senders finished with this slot */
} ti_slot;
typedef struct {
    node *self;
    int P;
    const int *pa_of, *pb_of;
    ti_slot slot[kTiRing];
    size_t ready;              /* batches published so far */
    size_t total;              /* number of batches, known from the start */
    int failed;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ti_ring;
typedef struct { ti_ring *ring; int owner; } ti_sender;
static void *ti_sender_main(void *arg) {
    ti_sender *t = arg;
    ti_ring *R = t->ring;
    for (size_t b = 0; b < R->total; b++) {
        pthread_mutex_lock(&R->mu);
        while (R->ready <= b && !R->failed) pthread_cond_wait(&R->cv, &R->mu);
        int failed = R->failed;
        pthread_mutex_unlock(&R->mu);
        if (failed) break;
        ti_slot *S = &R->slot[b % kTiRing];
        const int *pa = R->pa_of + S->q0, *pb = R->pb_of + S->q0;
        int bad = 0;
        for (size_t q = 0; q < S->nb && !bad; q++) {
            if (pa[q] == t->owner) bad |= net_send_flush(R->self, t->owner + 1, S->frames[2 * q].buf, S->frames[2 * q].len);
            if (pb[q] == t->owner) bad |= net_send_flush(R->self, t->owner + 1, S->frames[2 * q + 1].buf, S->frames[2 * q + 1].len);
        }
        pthread_mutex_lock(&R->mu);
        if (bad) R->failed = 1;
        S->done++;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->mu);
        if (bad) break;
    }
    return NULL;
}
/* Messages of ~0.5 MB are allocated and freed hundreds of thousands of times, by different threads:
 * keep them on the heap instead of one mmap/munmap (page faults, TLB shootdowns) per message. */
static void tune_malloc(void) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
}

/* ---------------------------------------------------------------- TI mode on one node: --ti_ring
 * All parties of phase 1 share a node: the vectors of inner_product_ti never leave HBM.  The trusted
 * initializer writes x / y of every cross-party pair straight into a device ring of the data provider that
 * is entitled to it (one ring PER provider: party a never maps b's x), and two providers exchange b + x and
 * a - y through one-slot device rings they map from each other (hipIpc).  The sockets carry one-byte tokens
 * and the scalars (r, <x,y> - r).  Same values, same pair order, same shares as the socket protocol
 * (src/phase1.c:148-339); what does not exist here is the byte stream.
 *   batch t = cross pairs [t B, (t + 1) B) of the loop order; every party derives the same partition.
 *   TI  : wait for a free slot of every provider that has entries in t (3 slots, acks 'K'), generate + scatter,
 *         send 'T' + that provider's scalars
 *   DP  : per batch with entries: wait 'T'; as party b (towards higher parties): masks -> own ring, token 'M';
 *         as party a (towards lower parties): wait 'M', fused step -> replies in own ring + shares, token 'A';
 *         as party b again: wait 'A', shares; ack 'K'.  One thread per process, no cycle in the waits. */
static int g_ti_ring = 0;
void protocol_set_ti_ring(int on) { g_ti_ring = on; }
typedef struct { int pa, pb; uint32_t ci, cj; } xpair;
static size_t enumerate_cross(config *c, xpair **out) {
    size_t cap = 0, np = 0;
    xpair *v = NULL;
    for (size_t i = 0; i <= c->d; i++)
        for (size_t j = 0; j <= i && j < c->d; j++) {
            int pa = config_owner(c, i), pb = config_owner(c, j);
            if (pa == pb) continue;
            if (np == cap) { cap = cap ? 2 * cap : 1024; v = realloc(v, cap * sizeof *v); if (!v) return 0; }
            xpair x = {pa, pb, (uint32_t)i, (uint32_t)j};
            v[np++] = x;
        }
    *out = v;
    return np;
}
enum { kTiRingSlots = 3 };
/* pairs per batch: a batch costs every party a fixed ~0.5 ms of tokens and device synchronisations whatever its size, and
 * config 4 has 1e5 pairs of 5e4 words -- with 64 MiB slots (167 pairs, 602 batches; rounds 2-3) its phase 1 was 0.55 s of
 * which two thirds were those fixed costs.  LINREG_TI_SLOT_MB overrides the slot size (experiments). */
static size_t ti_ring_batch(size_t n) {
    size_t slot_mb = 256;
    const char *e = getenv("LINREG_TI_SLOT_MB");
    if (e && atoi(e) > 0) slot_mb = (size_t)atoi(e);
    size_t b = (slot_mb << 20) / (n * 8);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
}
static int tok_send(node *self, int to, char t) { return net_send(self, to, &t, 1); }
static int tok_expect(node *self, int from, char want) {
    char t = 0;
    if (net_recv(self, from, &t, 1) || t != want) { fprintf(stderr, "ring protocol: expected '%c' from party %d\n", want, from); return 1; }
    return 0;
}

static int run_trusted_initializer_ring(node *self, config *c, int w1, int device, const uint8_t seed[16]) {
    const size_t n = c->n;
    const int NP = c->num_parties;
    xpair *xp = NULL;
    const size_t np = enumerate_cross(c, &xp), B = ti_ring_batch(n), slotb = B * n * 8;
    void *ring[64] = {0};
    size_t issued[64] = {0}, acked[64] = {0}, cnt[64];
    uint64_t *scal[64] = {0};
    void **xdst = malloc(B * sizeof(void *)), **ydst = malloc(B * sizeof(void *));
    uint64_t *r = malloc(B * 8), *xyr = malloc(B * 8);
    uint8_t *msg = malloc(1 + B * 8);
    int rc = 1;
    if ((np && !xp) || !xdst || !ydst || !r || !xyr || !msg || NP > 64) goto out;
    for (int k = 2; k < NP; k++) {
        uint8_t h[64];
        if (lgc_dev_alloc(device, kTiRingSlots * slotb, &ring[k], h)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
        scal[k] = malloc(B * 8);
        if (!scal[k] || send_blob(self, k + 1, h, 64)) goto out;
    }
    for (int k = 2; k < NP; k++) if (tok_expect(self, k + 1, 'O')) goto out;   /* every provider has mapped its ring */
    for (size_t q0 = 0; q0 < np; q0 += B) {
        const size_t nb = np - q0 < B ? np - q0 : B;
        int has[64] = {0};
        for (size_t q = 0; q < nb; q++) { has[xp[q0 + q].pa] = 1; has[xp[q0 + q].pb] = 1; }
        for (int k = 2; k < NP; k++) {                       /* a free slot for everybody involved */
            cnt[k] = 0;
            while (has[k] && issued[k] - acked[k] >= kTiRingSlots) { if (tok_expect(self, k + 1, 'K')) goto out; acked[k]++; }
        }
        for (size_t q = 0; q < nb; q++) {
            const xpair *x = &xp[q0 + q];
            ydst[q] = (char *)ring[x->pa] + ((issued[x->pa] % kTiRingSlots) * B + cnt[x->pa]) * n * 8;   /* a: (y, <x,y> - r) */
            xdst[q] = (char *)ring[x->pb] + ((issued[x->pb] % kTiRingSlots) * B + cnt[x->pb]) * n * 8;   /* b: (x, r) */
            cnt[x->pa]++; cnt[x->pb]++;
        }
        if (lgc_ti_generate_scatter(device, seed, q0, nb, n, w1, xdst, ydst, r, xyr)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
        for (int k = 2; k < NP; k++) cnt[k] = 0;
        for (size_t q = 0; q < nb; q++) { scal[xp[q0 + q].pa][cnt[xp[q0 + q].pa]++] = xyr[q]; scal[xp[q0 + q].pb][cnt[xp[q0 + q].pb]++] = r[q]; }
        for (int k = 2; k < NP; k++) {
            if (!has[k]) continue;
            msg[0] = 'T';
            memcpy(msg + 1, scal[k], cnt[k] * 8);
            if (net_send(self, k + 1, msg, 1 + cnt[k] * 8)) goto out;
            issued[k]++;
        }
    }
    for (int k = 2; k < NP; k++) while (acked[k] < issued[k]) { if (tok_expect(self, k + 1, 'K')) goto out; acked[k]++; }
    rc = 0;
out:
    for (int k = 2; k < NP && k < 64; k++) { lgc_dev_free(ring[k]); free(scal[k]); }
    free(xp); free(xdst); free(ydst); free(r); free(xyr); free(msg);
    return rc;
}

static int run_party_ti_ring(node *self, config *c, lgc_p1 *p1, int device, uint64_t *share_A, uint64_t *share_b) {
    const size_t n = c->n, d = c->d;
    const int NP = c->num_parties, me = c->party - 1;
    xpair *xp = NULL;
    const size_t np = enumerate_cross(c, &xp), B = ti_ring_batch(n), slotb = B * n * 8;
    void *ti = NULL, *mine[64] = {0}, *theirs[64] = {0};
    int shared[64] = {0};
    uint32_t *col = malloc(B * sizeof *col);
    int *peer = malloc(B * sizeof *peer);
    uint64_t **dst = malloc(B * sizeof *dst), *scal = malloc(B * 8 + 8), *shares = malloc(B * 8 + 8);
    uint8_t *msg = malloc(1 + B * 8);
    int rc = 1;
    if ((np && !xp) || !col || !peer || !dst || !scal || !shares || !msg || NP > 64) goto out;
    {
        uint8_t h[64];
        if (recv_blob(self, 1, h, 64) || lgc_dev_open(device, h, &ti)) { fprintf(stderr, "could not map the TI ring: %s\n", lgc_last_error()); goto out; }
        if (tok_send(self, 1, 'O')) goto out;
    }
    for (size_t q = 0; q < np; q++) { if (xp[q].pa == me) shared[xp[q].pb] = 1; if (xp[q].pb == me) shared[xp[q].pa] = 1; }
    for (int k = 2; k < NP; k++) {
        if (!shared[k]) continue;
        uint8_t hm[64], ht[64];
        if (lgc_dev_alloc(device, slotb, &mine[k], hm)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
        if (send_blob(self, k + 1, hm, 64) || recv_blob(self, k + 1, ht, 64) || lgc_dev_open(device, ht, &theirs[k])) {
            fprintf(stderr, "could not exchange ring handles with party %d: %s\n", k + 1, lgc_last_error()); goto out;
        }
    }
    if (lgc_p1_set_device_io(p1, 1)) goto out;
    size_t m = 0;
    for (size_t q0 = 0; q0 < np; q0 += B) {
        const size_t nb = np - q0 < B ? np - q0 : B;
        size_t cnt = 0;
        int has[64] = {0};
        for (size_t q = 0; q < nb; q++) {
            const xpair *x = &xp[q0 + q];
            if (x->pa != me && x->pb != me) continue;
            const int is_a = x->pa == me;
            peer[cnt] = is_a ? x->pb : x->pa;
            col[cnt] = is_a ? x->ci : x->cj;
            dst[cnt] = x->ci < d ? share_A + idx(x->ci, x->cj) : share_b + x->cj;
            has[peer[cnt]] = 1;
            cnt++;
        }
        if (!cnt) continue;
        if (net_recv(self, 1, msg, 1 + cnt * 8) || msg[0] != 'T') { fprintf(stderr, "ring protocol: no batch from the TI\n"); goto out; }
        memcpy(scal, msg + 1, cnt * 8);
        char *base = (char *)ti + (m % kTiRingSlots) * slotb;
        /* party b towards the higher parties: b + x */
        for (int k = me + 1; k < NP; k++) {
            if (!has[k]) continue;
            size_t pos = 0;
            for (size_t e = 0; e < cnt;) {
                if (peer[e] != k) { e++; continue; }
                size_t len = 1;
                while (e + len < cnt && peer[e + len] == k) len++;
                if (lgc_p1_mask(p1, col + e, len, (const uint64_t *)(base + e * n * 8), +1, (uint64_t *)((char *)mine[k] + pos * n * 8))) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
                pos += len; e += len;
            }
            if (tok_send(self, k + 1, 'M')) goto out;
        }
        /* party a towards the lower parties: a - y and <b + x, y> - (<x,y> - r) */
        for (int k = 2; k < me; k++) {
            if (!has[k]) continue;
            if (tok_expect(self, k + 1, 'M')) goto out;
            size_t pos = 0;
            for (size_t e = 0; e < cnt;) {
                if (peer[e] != k) { e++; continue; }
                size_t len = 1;
                while (e + len < cnt && peer[e + len] == k) len++;
                if (lgc_p1_ti_a_batch(p1, col + e, len, (const uint64_t *)(base + e * n * 8), (const uint64_t *)((char *)theirs[k] + pos * n * 8),
                                      scal + e, (uint64_t *)((char *)mine[k] + pos * n * 8), shares)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
                for (size_t i = 0; i < len; i++) *dst[e + i] = shares[i];
                pos += len; e += len;
            }
            if (tok_send(self, k + 1, 'A')) goto out;
        }
        /* party b again: <a - y, b> - r */
        for (int k = me + 1; k < NP; k++) {
            if (!has[k]) continue;
            if (tok_expect(self, k + 1, 'A')) goto out;
            size_t pos = 0;
            for (size_t e = 0; e < cnt;) {
                if (peer[e] != k) { e++; continue; }
                size_t len = 1;
                while (e + len < cnt && peer[e + len] == k) len++;
                if (lgc_p1_dot(p1, (const uint64_t *)((char *)theirs[k] + pos * n * 8), NULL, col + e, len, scal + e, shares)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
                for (size_t i = 0; i < len; i++) *dst[e + i] = shares[i];
                pos += len; e += len;
            }
        }
        if (tok_send(self, 1, 'K')) goto out;
        m++;
    }
    rc = 0;
out:
    (void)lgc_p1_set_device_io(p1, 0);
    for (int k = 2; k < NP && k < 64; k++) { if (theirs[k]) lgc_dev_close(theirs[k]); lgc_dev_free(mine[k]); }
    if (ti) lgc_dev_close(ti);
    free(xp); free(col); free(peer); free(dst); free(scal); free(shares); free(msg);
    return rc;
}

int run_trusted_initializer(node *self, config *c, int w1, int device) {
    tune_malloc();
    pmsg_set_limit(c->n);
    uint8_t seed[16];
    if (RAND_bytes(seed, sizeof seed) != 1) { fprintf(stderr, "RAND_bytes failed\n"); return 1; }   /* newBCipherRandomGen (src/phase1.c:243) */
#ifdef LINREG_TEST_HOOKS
    /* bin/linreg_testhooks only (share-level parity tests): 32 hex digits pin the TI stream.  The
     * production binaries are built without this: whoever sets the TI's seed knows every mask. */
    const char *fixed = getenv("LINREG_TI_SEED");
    if (fixed && strlen(fixed) == 32)
        for (int i = 0; i < 16; i++) { unsigned v = 0; sscanf(fixed + 2 * i, "%2x", &v); seed[i] = (uint8_t)v; }
#endif
    if (g_ti_ring) return run_trusted_initializer_ring(self, c, w1, device, seed);
    const size_t n = c->n;
    /* enumerate the cross-party pairs in the loop order of src/phase1.c:256-258, then generate the
     * randomness in batches on the GPU and send the two messages of every pair in that order */
    size_t cap = 0, np = 0;
    int *pa_of = NULL, *pb_of = NULL;
    ti_sender *snd = NULL;
    pthread_t *tid = NULL;
    ti_ring *R = NULL;
    uint64_t *x = NULL, *y = NULL, *r = NULL, *xyr = NULL;
    int rc = 1, started = 0;
    for (size_t i = 0; i <= c->d; i++)
        for (size_t j = 0; j <= i && j < c->d; j++) {
            int pa = config_owner(c, i), pb = config_owner(c, j);
            if (pa == pb) continue;
            if (np == cap) { cap = cap ? 2 * cap : 1024; pa_of = realloc(pa_of, cap * sizeof(int)); pb_of = realloc(pb_of, cap * sizeof(int)); }
            pa_of[np] = pa; pb_of[np] = pb; np++;
        }
    size_t batch = ((size_t)64 << 20) / (n * 8);          /* about 64 MiB of x (and of y) per batch */
    if (batch < 1) batch = 1;
    if (batch > 1024) batch = 1024;
    x = malloc(batch * n * 8); y = malloc(batch * n * 8); r = malloc(batch * 8); xyr = malloc(batch * 8);
    const int P = c->num_parties - 2;
    snd = calloc((size_t)P, sizeof *snd);
    tid = calloc((size_t)P, sizeof *tid);
    R = calloc(1, sizeof *R);
    check(x && y && r && xyr && snd && tid && R, "out of memory");
    R->self = self; R->P = P; R->pa_of = pa_of; R->pb_of = pb_of;
    R->total = (np + batch - 1) / batch;
    pthread_mutex_init(&R->mu, NULL); pthread_cond_init(&R->cv, NULL);
    for (int k = 0; k < kTiRing; k++) {
        R->slot[k].frames = calloc(2 * batch, sizeof(ti_frame));
        check(R->slot[k].frames, "out of memory");
        R->slot[k].done = P;                              /* free */
    }
    for (int k = 0; k < P; k++) {
        snd[k].ring = R; snd[k].owner = k + 2;
        check(!pthread_create(&tid[k], NULL, ti_sender_main, &snd[k]), "pthread_create failed");
        started = k + 1;
    }
    enum { kEnc = 16 };
    ti_encoder enc[kEnc];
    pthread_t etid[kEnc];
    const int timing = getenv("LINREG_TIMING") != NULL;
    double t_gen = 0, t_enc = 0, t_wait = 0, t_all = wall_clock();
    for (size_t b = 0; b < R->total; b++) {
        const size_t q0 = b * batch, nb = np - q0 < batch ? np - q0 : batch;
        ti_slot *S = &R->slot[b % kTiRing];
        double t0 = wall_clock();
        pthread_mutex_lock(&R->mu);                       /* wait until every sender is through with the slot */
        while (S->done < P && !R->failed) pthread_cond_wait(&R->cv, &R->mu);
        int failed = R->failed;
        pthread_mutex_unlock(&R->mu);
        check(!failed, "Could not send message to a data provider");
        for (size_t m = 0; m < 2 * batch; m++) { free(S->frames[m].buf); S->frames[m].buf = NULL; }
        double t1 = wall_clock();
        LGC(lgc_ti_generate(device, seed, q0, nb, n, w1, x, y, r, xyr));
        double t2 = wall_clock();
        for (int e = 0; e < kEnc; e++) {
            ti_encoder t = {n, nb, (size_t)e, (size_t)kEnc, x, y, r, xyr, S->frames, 0};
            enc[e] = t;
            check(!pthread_create(&etid[e], NULL, ti_encoder_main, &enc[e]), "pthread_create failed");
        }
        for (int e = 0; e < kEnc; e++) { pthread_join(etid[e], NULL); failed |= enc[e].failed; }
        check(!failed, "out of memory while encoding TI messages");
        double t3 = wall_clock();
        t_wait += t1 - t0; t_gen += t2 - t1; t_enc += t3 - t2;
        pthread_mutex_lock(&R->mu);
        S->q0 = q0; S->nb = nb; S->done = 0;
        R->ready = b + 1;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->mu);
    }
    for (int k = 0; k < P; k++) pthread_join(tid[k], NULL);
    started = 0;
    check(!R->failed, "Could not send message to a data provider");
    if (timing) fprintf(stderr, "TI: %zu pairs, batches of %zu, ring of %d: generate %.2fs, encode %.2fs, waiting for a free slot %.2fs, total %.2fs\n",
                        np, batch, (int)kTiRing, t_gen, t_enc, t_wait, wall_clock() - t_all);
    rc = 0;
error:
    if (R) {
        if (started) {                                    /* unblock and collect the senders */
            pthread_mutex_lock(&R->mu); R->failed = 1; pthread_cond_broadcast(&R->cv); pthread_mutex_unlock(&R->mu);
            for (int k = 0; k < started; k++) pthread_join(tid[k], NULL);
        }
        for (int k = 0; k < kTiRing; k++) {
            if (R->slot[k].frames) for (size_t m = 0; m < 2 * batch; m++) free(R->slot[k].frames[m].buf);
            free(R->slot[k].frames);
        }
        free(R);
    }
    free(x); free(y); free(r); free(xyr); free(pa_of); free(pb_of); free(snd); free(tid);
    return rc;
}

/* ---------------------------------------------------------------- phase 1: data provider */
/* TI-mode plumbing: a bounded queue of decoded TI messages per peer, and the per-peer worker */
typedef struct { uint8_t *raw; size_t len; uint64_t *vec; uint64_t val; } ti_item;   /* raw: undecoded message (decoded by the worker) */
typedef struct {
    ti_item *items;
    size_t cap, head, count;
    int closed;                 /* no more pushes (reader done) or no more pops (worker failed) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ti_queue;
static void ti_queue_init(ti_queue *q, size_t cap) {
    q->items = calloc(cap, sizeof *q->items); q->cap = cap; q->head = q->count = 0; q->closed = 0;
    pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL);
}
static void ti_queue_destroy(ti_queue *q) {
    for (size_t i = 0; i < q->count; i++) { free(q->items[(q->head + i) % q->cap].vec); free(q->items[(q->head + i) % q->cap].raw); }
    free(q->items); pthread_mutex_destroy(&q->mu); pthread_cond_destroy(&q->cv);
}
static void ti_queue_close(ti_queue *q) {
    pthread_mutex_lock(&q->mu); q->closed = 1; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu);
}
static int ti_queue_push(ti_queue *q, ti_item it) {
    pthread_mutex_lock(&q->mu);
    while (q->count == q->cap && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
    if (q->closed) { pthread_mutex_unlock(&q->mu); return 1; }
    q->items[(q->head + q->count++) % q->cap] = it;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    return 0;
}
static int ti_queue_pop(ti_queue *q, ti_item *it) {
    pthread_mutex_lock(&q->mu);
    while (q->count == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
    if (q->count == 0) { pthread_mutex_unlock(&q->mu); return 1; }
    *it = q->items[q->head]; q->head = (q->head + 1) % q->cap; q->count--;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    return 0;
}
typedef struct { int peer; int is_a; uint32_t col; uint64_t *dst; } ti_pair;
typedef struct {
    node *self; lgc_p1 *p1; size_t n; int peer;
    const ti_pair *pairs; size_t npairs;
    ti_queue *q;
    int failed;
} ti_worker;
/* recv_pmsg from a peer data provider, timed like the reference's wait_total (src/phase1.c:177-183, 211-217) */
static int recv_pmsg_timed(node *self, int from, uint64_t **vec, size_t *n, uint64_t *value) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int rc = recv_pmsg(self, from, vec, n, value);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    self->wait_ns[from - 1] += (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
    return rc;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* non-blocking pop: 1 when the queue is empty right now */
static int ti_queue_try_pop(ti_queue *q, ti_item *it) {
    pthread_mutex_lock(&q->mu);
    if (q->count == 0) { pthread_mutex_unlock(&q->mu); return 1; }
    *it = q->items[q->head]; q->head = (q->head + 1) % q->cap; q->count--;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    return 0;
}
/* one message from a peer data provider, decoded straight into `dst` (n words, page-locked); raw / rawcap:
 * the caller's reusable receive buffer.  Timed like the reference's wait_total. */
static int recv_pmsg_into_timed(node *self, int from, uint8_t **raw, size_t *rawcap, uint64_t *dst, size_t n, uint64_t *value) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t sz = 0, got = 0;
    int rc = net_recv(self, from, &sz, sizeof sz) || sz > g_pmsg_limit;
    if (!rc && sz > *rawcap) { free(*raw); *raw = malloc(sz + sz / 8); *rawcap = *raw ? sz + sz / 8 : 0; rc = !*raw; }
    if (!rc) rc = net_recv(self, from, *raw, sz);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    self->wait_ns[from - 1] += (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
    if (!rc) rc = pmsg_unpack_into(*raw, sz, dst, n, &got, value) || got != n;
    return rc;
}
/* The per-peer workers batch: a run of pairs with one peer is one device call (lgc_p1_mask / lgc_p1_dot /
 * lgc_p1_ti_a_batch over up to kTiBatch pairs) on page-locked buffers the messages are decoded into, instead
 * of one call per pair from pageable memory.  A batch is whatever has arrived (at least one pair), so
 * nothing waits for a batch to fill.  Same bytes, same order on every socket. */
enum { kTiBatch = 16, kTiSlots = 32 };

/* Party b of a run of pairs with one peer, pipelined.  b's first message (b + x) depends only on
 * the TI's message, so this thread sends the masks of successive pairs back to back, while a second
 * thread receives party a's replies and finishes the shares (<a - y, b> - r): party b then never
 * idles for a round trip, and party a always finds its next input waiting. */
typedef struct {
    ti_worker *w;
    const ti_pair **pr;        /* this worker's pairs, in order */
    uint64_t *r;               /* the TI's r of each pair (filled by the sender side) */
    size_t total;
    size_t sent;               /* pairs whose mask has been sent (published under mu) */
    int stop;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ti_b_pipe;
static void *ti_b_finisher(void *arg) {
    ti_b_pipe *bp = arg;
    ti_worker *w = bp->w;
    const size_t n = w->n;
    const int to = w->peer + 1;
    uint64_t *in = lgc_host_alloc(kTiBatch * n * 8);
    uint8_t *raw = NULL; size_t rawcap = 0;
    uint32_t cols[kTiBatch]; uint64_t shares[kTiBatch];
    if (!in) w->failed = 1;
    for (size_t k = 0; k < bp->total && !w->failed;) {
        pthread_mutex_lock(&bp->mu);
        while (bp->sent <= k && !bp->stop) pthread_cond_wait(&bp->cv, &bp->mu);
        size_t avail = bp->sent - k;
        pthread_mutex_unlock(&bp->mu);
        if (!avail) break;
        size_t nb = avail < kTiBatch ? avail : kTiBatch;
        for (size_t i = 0; i < nb && !w->failed; i++) {
            uint64_t inval = 0;
            cols[i] = bp->pr[k + i]->col;
            if (recv_pmsg_into_timed(w->self, to, &raw, &rawcap, in + i * n, n, &inval)) { fprintf(stderr, "Could not receive message from party A (%d)\n", w->peer); w->failed = 1; }
        }
        if (w->failed) break;
        if (lgc_p1_dot(w->p1, in, 0, cols, nb, bp->r + k, shares)) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; break; }   /* <a-y, b> - r */
        for (size_t i = 0; i < nb; i++) *bp->pr[k + i]->dst = shares[i];
        k += nb;
    }
    free(raw); lgc_host_free(in);
    return NULL;
}
static int ti_worker_b_pipelined(ti_worker *w, const ti_pair **mine, size_t total) {
    const size_t n = w->n;
    const int to = w->peer + 1;
    ti_b_pipe bp;
    memset(&bp, 0, sizeof bp);
    bp.w = w; bp.pr = mine; bp.total = total;
    bp.r = malloc((total + 1) * sizeof *bp.r);
    uint64_t *x = lgc_host_alloc(kTiBatch * n * 8), *m = lgc_host_alloc(kTiBatch * n * 8);
    uint32_t cols[kTiBatch];
    pthread_mutex_init(&bp.mu, NULL); pthread_cond_init(&bp.cv, NULL);
    pthread_t fin;
    int have_fin = bp.r && x && m && !pthread_create(&fin, NULL, ti_b_finisher, &bp);
    if (!have_fin) w->failed = 1;
    for (size_t k = 0; k < total && !w->failed;) {
        size_t nb = 0;
        while (nb < kTiBatch && k + nb < total) {
            ti_item it = {0, 0, 0, 0};
            size_t ti_n = 0;
            if (nb == 0 ? ti_queue_pop(w->q, &it) : ti_queue_try_pop(w->q, &it)) { if (nb == 0) w->failed = 1; break; }
            if (pmsg_unpack_into(it.raw, it.len, x + nb * n, n, &ti_n, &it.val) || ti_n != n) { fprintf(stderr, "Could not decode message from TI\n"); w->failed = 1; }
            free(it.raw);
            if (w->failed) break;
            cols[nb] = mine[k + nb]->col;
            bp.r[k + nb] = it.val;
            nb++;
        }
        if (w->failed || !nb) break;
        if (lgc_p1_mask(w->p1, cols, nb, x, +1, m)) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; break; }        /* b + x */
        for (size_t i = 0; i < nb && !w->failed; i++)
            if (send_pmsg(w->self, to, m + i * n, n, 0)) { fprintf(stderr, "Could not send message to party A (%d)\n", w->peer); w->failed = 1; }
        if (w->failed) break;
        k += nb;
        pthread_mutex_lock(&bp.mu); bp.sent = k; pthread_cond_broadcast(&bp.cv); pthread_mutex_unlock(&bp.mu);
    }
    pthread_mutex_lock(&bp.mu); bp.stop = 1; pthread_cond_broadcast(&bp.cv); pthread_mutex_unlock(&bp.mu);
    if (have_fin) pthread_join(fin, NULL);
    pthread_mutex_destroy(&bp.mu); pthread_cond_destroy(&bp.cv);
    free(bp.r); lgc_host_free(x); lgc_host_free(m);
    return w->failed;
}

/* Party a of a run of pairs with one peer, as two stages: a prefetch thread takes the TI's message
 * and party b's message off the queue / socket and decodes both into a ring of page-locked slots, while
 * this thread runs the fused device step over the slots that are ready and sends the replies. */
typedef struct {
    ti_worker *w;
    size_t total;
    uint64_t *y, *in;          /* kTiSlots x n words each, page-locked */
    uint64_t sub[kTiSlots];
    size_t produced, consumed;
    int stop;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ti_a_pipe;
static void *ti_a_prefetch(void *arg) {
    ti_a_pipe *ap = arg;
    ti_worker *w = ap->w;
    const size_t n = w->n;
    const int to = w->peer + 1;
    uint8_t *raw = NULL; size_t rawcap = 0;
    for (size_t k = 0; k < ap->total; k++) {
        pthread_mutex_lock(&ap->mu);
        while (ap->produced - ap->consumed == kTiSlots && !ap->stop) pthread_cond_wait(&ap->cv, &ap->mu);
        int stop = ap->stop;
        pthread_mutex_unlock(&ap->mu);
        if (stop) break;
        const size_t slot = k % kTiSlots;
        ti_item it = {0, 0, 0, 0};
        size_t ti_n = 0;
        uint64_t inval = 0;
        int bad = 0;
        if (ti_queue_pop(w->q, &it)) bad = 1;
        else if (pmsg_unpack_into(it.raw, it.len, ap->y + slot * n, n, &ti_n, &it.val) || ti_n != n) { fprintf(stderr, "Could not decode message from TI\n"); bad = 1; }
        else if (recv_pmsg_into_timed(w->self, to, &raw, &rawcap, ap->in + slot * n, n, &inval)) { fprintf(stderr, "Could not receive message from party B (%d)\n", w->peer); bad = 1; }
        free(it.raw);
        pthread_mutex_lock(&ap->mu);
        if (bad) { ap->stop = 1; w->failed = 1; }
        else { ap->sub[slot] = it.val; ap->produced++; }
        pthread_cond_broadcast(&ap->cv);
        pthread_mutex_unlock(&ap->mu);
        if (bad) break;
    }
    free(raw);
    return NULL;
}
static int ti_worker_a_pipelined(ti_worker *w, const ti_pair **mine, size_t total) {
    const size_t n = w->n;
    const int to = w->peer + 1;
    ti_a_pipe ap;
    memset(&ap, 0, sizeof ap);
    ap.w = w; ap.total = total;
    ap.y = lgc_host_alloc((size_t)kTiSlots * n * 8); ap.in = lgc_host_alloc((size_t)kTiSlots * n * 8);
    uint64_t *out = lgc_host_alloc((size_t)kTiBatch * n * 8);
    uint32_t cols[kTiBatch]; uint64_t shares[kTiBatch], sub[kTiBatch];
    pthread_mutex_init(&ap.mu, NULL); pthread_cond_init(&ap.cv, NULL);
    pthread_t pre;
    int have = ap.y && ap.in && out && !pthread_create(&pre, NULL, ti_a_prefetch, &ap);
    if (!have) w->failed = 1;
    for (size_t k = 0; k < total && !w->failed;) {
        pthread_mutex_lock(&ap.mu);
        while (ap.produced == ap.consumed && !ap.stop) pthread_cond_wait(&ap.cv, &ap.mu);
        size_t avail = ap.produced - ap.consumed;
        pthread_mutex_unlock(&ap.mu);
        if (!avail) { w->failed = 1; break; }
        const size_t slot = k % kTiSlots;
        size_t nb = avail < kTiBatch ? avail : kTiBatch;
        if (nb > kTiSlots - slot) nb = kTiSlots - slot;          /* a batch is contiguous in the ring */
        for (size_t i = 0; i < nb; i++) { cols[i] = mine[k + i]->col; sub[i] = ap.sub[slot + i]; }
        /* a - y and <b+x, y> - (xy - r) for the whole batch */
        if (lgc_p1_ti_a_batch(w->p1, cols, nb, ap.y + slot * n, ap.in + slot * n, sub, out, shares)) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; break; }
        for (size_t i = 0; i < nb && !w->failed; i++)
            if (send_pmsg(w->self, to, out + i * n, n, 0)) { fprintf(stderr, "Could not send message to party B (%d)\n", w->peer); w->failed = 1; }
        if (w->failed) break;
        for (size_t i = 0; i < nb; i++) *mine[k + i]->dst = shares[i];
        k += nb;
        pthread_mutex_lock(&ap.mu); ap.consumed = k; pthread_cond_broadcast(&ap.cv); pthread_mutex_unlock(&ap.mu);
    }
    pthread_mutex_lock(&ap.mu); ap.stop = 1; pthread_cond_broadcast(&ap.cv); pthread_mutex_unlock(&ap.mu);
    if (w->failed) ti_queue_close(w->q);                         /* the prefetch thread may be blocked in a pop */
    if (have) pthread_join(pre, NULL);
    pthread_mutex_destroy(&ap.mu); pthread_cond_destroy(&ap.cv);
    lgc_host_free(ap.y); lgc_host_free(ap.in); lgc_host_free(out);
    return w->failed;
}

static void *ti_worker_main(void *arg) {
    ti_worker *w = arg;
    const int timing = getenv("LINREG_TIMING") != NULL;
    double t_pop = 0, t_recv = 0, t_gpu = 0, t_send = 0, t0 = 0;
    size_t done = 0;
    const size_t n = w->n;
    const int to = w->peer + 1;
    {   /* the role towards one peer is fixed by the column ownership (the later party owns the rows):
         * run the pipelined form of that role */
        size_t cnt = 0, as_b = 0;
        for (size_t k = 0; k < w->npairs; k++) if (w->pairs[k].peer == w->peer) { cnt++; as_b += !w->pairs[k].is_a; }
        if (cnt && (as_b == cnt || as_b == 0) && !getenv("LINREG_TI_LOCKSTEP")) {
            const ti_pair **mine = malloc(cnt * sizeof *mine);
            size_t m = 0;
            for (size_t k = 0; k < w->npairs; k++) if (w->pairs[k].peer == w->peer) mine[m++] = &w->pairs[k];
            if (as_b) ti_worker_b_pipelined(w, mine, cnt); else ti_worker_a_pipelined(w, mine, cnt);
            free(mine);
            if (w->failed) ti_queue_close(w->q);
            return NULL;
        }
    }
    uint64_t *tmp = malloc(n * 8);
    for (size_t k = 0; k < w->npairs && !w->failed; k++) {
        const ti_pair *pr = &w->pairs[k];
        if (pr->peer != w->peer) continue;
        ti_item it = {0, 0, 0, 0};
        uint64_t *in = 0, inval = 0, share = 0, sub;
        size_t in_n = 0, ti_n = 0;
        if (timing) t0 = now_s();
        if (ti_queue_pop(w->q, &it)) { w->failed = 1; break; }
        if (timing) t_pop += now_s() - t0;
        if (pmsg_unpack(it.raw, it.len, &it.vec, &ti_n, &it.val) || ti_n != n) {
            fprintf(stderr, "Could not decode message from TI\n"); w->failed = 1; free(it.raw); free(it.vec); break;
        }
        free(it.raw); it.raw = 0;
        uint32_t col = pr->col;
        sub = it.val;
        if (pr->is_a) {                                   /* party a (phase1.c:171-197) */
            double ta = timing ? now_s() : 0, tb, tc;
            if (recv_pmsg_timed(w->self, to, &in, &in_n, &inval) || in_n != n) { fprintf(stderr, "Could not receive message from party B (%d)\n", w->peer); w->failed = 1; }
            else if ((tb = timing ? now_s() : 0, lgc_p1_ti_a(w->p1, col, it.vec, in, sub, tmp, &share))) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; }   /* a - y and <b+x, y> - (xy - r) */
            else if ((tc = timing ? now_s() : 0, send_pmsg(w->self, to, tmp, n, 0))) { fprintf(stderr, "Could not send message to party B (%d)\n", w->peer); w->failed = 1; }
            else if (timing) { double td = now_s(); t_recv += tb - ta; t_gpu += tc - tb; t_send += td - tc; }
        } else {                                          /* party b (phase1.c:198-223) */
            if (lgc_p1_mask(w->p1, &col, 1, it.vec, +1, tmp)) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; }        /* b + x */
            else if (send_pmsg(w->self, to, tmp, n, 0)) { fprintf(stderr, "Could not send message to party A (%d)\n", w->peer); w->failed = 1; }
            else if (recv_pmsg_timed(w->self, to, &in, &in_n, &inval) || in_n != n) { fprintf(stderr, "Could not receive message from party A (%d)\n", w->peer); w->failed = 1; }
            else if (lgc_p1_dot(w->p1, in, 0, &col, 1, &sub, &share)) { fprintf(stderr, "%s\n", lgc_last_error()); w->failed = 1; }   /* <a-y, b> - r */
        }
        free(in); free(it.vec);
        if (!w->failed) *pr->dst = share;
        done++;
    }
    if (timing) fprintf(stderr, "worker peer %d: %zu pairs; as party a: recv %.2fs gpu %.2fs send %.2fs; TI queue wait %.2fs (all roles)\n",
                        w->peer, done, t_recv, t_gpu, t_send, t_pop);
    free(tmp);
    if (w->failed) ti_queue_close(w->q);                  /* unblock the reader */
    return NULL;
}

static void column_of(const int64_t *Xq, const int64_t *yq, size_t n, size_t d, size_t row, uint64_t *out) {
    for (size_t k = 0; k < n; k++) out[k] = (uint64_t)(row < d ? Xq[k * d + row] : yq[k]);
}

/* OT mode, sender side: u arrives on one helper thread and y leaves on another while the main
 * thread runs the extension + Gilboa kernels, two buffers each (the socket copies of 24 bytes per OT
 * are the cost of this phase; this overlaps the two directions and the GPU) */
typedef struct {
    node *self; int peer;
    size_t n; int w1; size_t npairs, per;
    uint8_t *u[2]; uint64_t *y[2];
    size_t recvd, gpu_done, sent;
    int failed;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ot_send_ctx;
static void *ot_send_recv_u(void *arg) {
    ot_send_ctx *c = arg;
    for (size_t q0 = 0, k = 0; q0 < c->npairs; q0 += c->per, k++) {
        size_t nb = c->npairs - q0 < c->per ? c->npairs - q0 : c->per;
        const uint64_t m = (uint64_t)nb * c->n * (uint64_t)c->w1;
        pthread_mutex_lock(&c->mu);
        while (k >= c->gpu_done + 2 && !c->failed) pthread_cond_wait(&c->cv, &c->mu);
        int bad = c->failed;
        pthread_mutex_unlock(&c->mu);
        if (bad) break;
        bad = recv_blob(c->self, c->peer, c->u[k & 1], lgc_ot_u_bytes(m));
        if (bad) fprintf(stderr, "OT: could not receive u\n");
        pthread_mutex_lock(&c->mu); if (bad) c->failed = 1; else c->recvd = k + 1; pthread_cond_broadcast(&c->cv); pthread_mutex_unlock(&c->mu);
        if (bad) break;
    }
    return NULL;
}
static void *ot_send_send_y(void *arg) {
    ot_send_ctx *c = arg;
    for (size_t q0 = 0, k = 0; q0 < c->npairs; q0 += c->per, k++) {
        size_t nb = c->npairs - q0 < c->per ? c->npairs - q0 : c->per;
        const uint64_t m = (uint64_t)nb * c->n * (uint64_t)c->w1;
        pthread_mutex_lock(&c->mu);
        while (c->gpu_done <= k && !c->failed) pthread_cond_wait(&c->cv, &c->mu);
        int bad = c->failed;
        pthread_mutex_unlock(&c->mu);
        if (bad) break;
        bad = send_blob(c->self, c->peer, c->y[k & 1], m * 8);
        if (bad) fprintf(stderr, "OT: could not send y\n");
        pthread_mutex_lock(&c->mu); if (bad) c->failed = 1; else c->sent = k + 1; pthread_cond_broadcast(&c->cv); pthread_mutex_unlock(&c->mu);
        if (bad) break;
    }
    return NULL;
}

/* OT mode, receiver side: the helper thread that starts batches (OT extension on the GPU) and sends
 * their u; at most two batches ahead of the finishing thread */
typedef struct {
    node *self; int to; lgc_ot_receiver *R;
    const int64_t *Xq, *yq; size_t n, d; int w1;
    const size_t *rows; size_t npairs, per;
    size_t started, finished;
    int failed;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ot_recv_ctx;
static void *ot_recv_starter(void *arg) {
    ot_recv_ctx *c = arg;
    const size_t n = c->n;
    const uint64_t mmax = (uint64_t)c->per * n * (uint64_t)c->w1;
    uint64_t *vals = lgc_host_alloc(c->per * n * 8);
    uint8_t *u = lgc_host_alloc(lgc_ot_u_bytes(mmax));
    int bad = !vals || !u;
    double tt[3] = {0, 0, 0};
    for (size_t q0 = 0, k = 0; q0 < c->npairs && !bad; q0 += c->per, k++) {
        size_t nb = c->npairs - q0 < c->per ? c->npairs - q0 : c->per;
        const uint64_t m = (uint64_t)nb * n * (uint64_t)c->w1;
        pthread_mutex_lock(&c->mu);
        while (k >= c->finished + 2 && !c->failed) pthread_cond_wait(&c->cv, &c->mu);
        bad = c->failed;
        pthread_mutex_unlock(&c->mu);
        if (bad) break;
        double t0 = wall_clock();
        for (size_t q = 0; q < nb; q++) column_of(c->Xq, c->yq, n, c->d, c->rows[q0 + q], vals + q * n);
        double t1 = wall_clock(), t2 = t1;
        if (lgc_ot_gilboa_recv_start(c->R, vals, nb, n, c->w1, u)) { fprintf(stderr, "%s\n", lgc_last_error()); bad = 1; }
        else if ((t2 = wall_clock(), send_blob(c->self, c->to, u, lgc_ot_u_bytes(m)))) { fprintf(stderr, "OT: could not send u\n"); bad = 1; }
        tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += wall_clock() - t2;
        pthread_mutex_lock(&c->mu); if (bad) c->failed = 1; else c->started = k + 1; pthread_cond_broadcast(&c->cv); pthread_mutex_unlock(&c->mu);
    }
    if (bad) { pthread_mutex_lock(&c->mu); c->failed = 1; pthread_cond_broadcast(&c->cv); pthread_mutex_unlock(&c->mu); }
    if (getenv("LINREG_TIMING")) fprintf(stderr, "OT receiver (start thread): columns %.2fs, gpu %.2fs, send u %.2fs\n", tt[0], tt[1], tt[2]);
    lgc_host_free(vals); lgc_host_free(u);
    return NULL;
}

/* OT mode with both data providers on one node (--ot_ring): u and y of the OT extension stay in HBM.  The
 * receiver owns a two-slot device ring for u, the sender one for y; each maps the other's through hipIpc
 * and only one-byte tokens cross the socket: receiver -> sender 'U' (u of the next batch is complete) and
 * 'A' (batch finished: its y slot and u slot are free again), sender -> receiver 'Y'.  Two batches in
 * flight.  Same OT transcripts as the socket path (the bytes just do not travel). */
static int ot_ring_token_send(node *self, int to, char t) { return net_send(self, to, &t, 1); }
static int ot_ring_token_recv(node *self, int from, char *t) { return net_recv(self, from, t, 1); }
static int ot_pair_ring(node *self, int peer_party, int i_am_sender, lgc_ot_sender *S, lgc_ot_receiver *R, int device,
                        const int64_t *Xq, const int64_t *yq, size_t n, size_t d, int w1,
                        const size_t *rows, size_t npairs, size_t per, uint64_t *shares) {
    const uint64_t mmax = (uint64_t)per * n * (uint64_t)w1;
    const size_t ub = lgc_ot_u_bytes(mmax), yb = (size_t)mmax * 8, vb = per * n * 8;
    const size_t nbatch = (npairs + per - 1) / per;
    void *mine = 0, *theirs = 0, *dvals = 0, *dsh = 0;
    uint8_t hmine[64], htheirs[64];
    uint64_t *vals = lgc_host_alloc(vb);
    int rc = 1;
    if (!vals) goto out;
    if (lgc_dev_alloc(device, 2 * (i_am_sender ? yb : ub), &mine, hmine) || lgc_dev_alloc(device, 2 * vb, &dvals, NULL) ||
        lgc_dev_alloc(device, per * 8 + 8, &dsh, NULL)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
    if (send_blob(self, peer_party, hmine, 64) || recv_blob(self, peer_party, htheirs, 64)) goto out;
    if (lgc_dev_open(device, htheirs, &theirs)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
    if (i_am_sender) {
        if (lgc_ot_sender_set_device_io(S, 1)) goto out;
        size_t have_u = 0, have_a = 0;
        for (size_t k = 0; k < nbatch; k++) {
            const size_t q0 = k * per, nb = npairs - q0 < per ? npairs - q0 : per;
            for (size_t q = 0; q < nb; q++) column_of(Xq, yq, n, d, rows[q0 + q], vals + q * n);
            if (lgc_dev_upload((char *)dvals + (k & 1) * vb, vals, nb * n * 8)) goto out;
            while (have_u <= k || (k >= 2 && have_a + 2 <= k)) {        /* u of batch k is there, y slot of batch k - 2 is free */
                char t = 0;
                if (ot_ring_token_recv(self, peer_party, &t)) goto out;
                if (t == 'U') have_u++; else if (t == 'A') have_a++; else goto out;
            }
            if (lgc_ot_gilboa_send(S, (const uint64_t *)((char *)dvals + (k & 1) * vb), nb, n, w1, (const uint8_t *)theirs + (k & 1) * ub,
                                   (uint64_t *)((char *)mine + (k & 1) * yb), (uint64_t *)dsh)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
            if (lgc_dev_download(shares + q0, dsh, nb * 8)) goto out;
            if (w1 == 32) for (size_t q = 0; q < nb; q++) shares[q0 + q] &= 0xffffffffull;
            if (ot_ring_token_send(self, peer_party, 'Y')) goto out;
        }
        while (have_a < nbatch) {                                        /* the receiver is done with every y slot */
            char t = 0;
            if (ot_ring_token_recv(self, peer_party, &t)) goto out;
            if (t == 'A') have_a++; else if (t != 'U') goto out;
        }
    } else {
        if (lgc_ot_receiver_set_device_io(R, 1)) goto out;
        size_t started = 0;
        for (size_t k = 0; k < nbatch + 2; k++) {
            if (k >= 2) {                                                /* finish batch k - 2 */
                const size_t f = k - 2, q0 = f * per, nb = npairs - q0 < per ? npairs - q0 : per;
                char t = 0;
                if (ot_ring_token_recv(self, peer_party, &t) || t != 'Y') goto out;
                if (lgc_ot_gilboa_recv_finish(R, (const uint64_t *)((char *)theirs + (f & 1) * yb), (uint64_t *)dsh)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
                if (lgc_dev_download(shares + q0, dsh, nb * 8)) goto out;
                if (w1 == 32) for (size_t q = 0; q < nb; q++) shares[q0 + q] &= 0xffffffffull;
                if (ot_ring_token_send(self, peer_party, 'A')) goto out;
            }
            if (started < nbatch) {                                      /* start the next batch: its slots are free now */
                const size_t q0 = started * per, nb = npairs - q0 < per ? npairs - q0 : per;
                for (size_t q = 0; q < nb; q++) column_of(Xq, yq, n, d, rows[q0 + q], vals + q * n);
                if (lgc_dev_upload((char *)dvals + (started & 1) * vb, vals, nb * n * 8)) goto out;
                if (lgc_ot_gilboa_recv_start(R, (const uint64_t *)((char *)dvals + (started & 1) * vb), nb, n, w1,
                                             (uint8_t *)mine + (started & 1) * ub)) { fprintf(stderr, "%s\n", lgc_last_error()); goto out; }
                if (ot_ring_token_send(self, peer_party, 'U')) goto out;
                started++;
            }
        }
    }
    rc = 0;
out:
    if (theirs) lgc_dev_close(theirs);
    lgc_dev_free(mine); lgc_dev_free(dvals); lgc_dev_free(dsh);
    lgc_host_free(vals);
    return rc;
}

int run_party(node *self, config *c, int precision, int precision_p2, int w1, int w2, int use_ot, int device,
                     uint64_t **res_A, uint64_t **res_b) {
    tune_malloc();
    pmsg_set_limit(c->n);
    const double t_start = wall_clock();
    const size_t n = c->n, d = c->d, T = d * (d + 1) / 2;
    const int me = c->party - 1, last = c->num_parties - 1;
    int64_t *Xq = malloc(n * d * 8), *yq = malloc(n * 8);
    uint64_t *share_A = calloc(T, 8), *share_b = calloc(d, 8), *va = 0, *vb = 0, *tmp = 0, *tmp2 = 0;
    lgc_p1 *p1 = 0;
    int rc = 1;
    double normalizer = sqrt(pow(2, precision) * (double)n);      /* src/phase1.c:473 */
    {
        const size_t oc0 = (size_t)c->index_owned[me], oc1 = me < last ? (size_t)c->index_owned[me + 1] : d;
        check(!read_own_columns(c->input, n, d, oc0, oc1, me == last, precision, normalizer, w2, Xq, yq), "Could not read data (dimensions or numbers invalid)");
    }
    if (getenv("LINREG_TIMING")) fprintf(stderr, "party %d: input parsed after %.2fs\n", c->party, wall_clock() - t_start);
    lgc_trace_mark("own columns parsed and quantised");
    LGC(lgc_p1_create(&p1, device, n, d, w1, precision));
    LGC(lgc_p1_set_data(p1, Xq, yq));
    lgc_trace_mark("phase-1 data on the device");
    const size_t c0 = (size_t)c->index_owned[me], c1 = me < last ? (size_t)c->index_owned[me + 1] : d;
    /* everything this party can do alone: its own block, incl. the floating-point diagonal */
    {
        size_t own = c1 - c0;
        uint64_t *blk = malloc((own * (own + 1) / 2 + 1) * 8), *bb = malloc((own + 1) * 8);
        LGC(lgc_p1_local(p1, c0, c1, me == last, blk, bb));
        for (size_t i = 0; i < own; i++) {
            for (size_t j = 0; j <= i; j++) share_A[idx(c0 + i, c0 + j)] = blk[i * (i + 1) / 2 + j];
            if (me == last) share_b[c0 + i] = bb[i];
        }
        free(blk); free(bb);
    }
    va = malloc(n * 8); vb = malloc(n * 8); tmp = malloc(n * 8); tmp2 = malloc(n * 8);
    if (!use_ot && g_ti_ring) {
        check(!run_party_ti_ring(self, c, p1, device, share_A, share_b), "TI-mode aggregation (device rings) failed");
        if (w1 == 32) { for (size_t k = 0; k < T; k++) share_A[k] &= 0xffffffffull; for (size_t k = 0; k < d; k++) share_b[k] &= 0xffffffffull; }
        if (getenv("LINREG_TIMING")) fprintf(stderr, "party %d: TI-mode aggregation (rings) done after %.2fs\n", c->party, wall_clock() - t_start);
    } else if (!use_ot) {
        /* TI mode.  The pairs are those of the loops at src/phase1.c:534-586 and every socket
         * carries its messages in that order (the TI socket: this party's pairs in loop order; a
         * peer socket: the pairs shared with that peer in loop order), so the byte streams are the
         * reference's.  Pairs with different peers are independent: one worker thread per peer runs
         * inner_product_ti for its pairs, fed by a reader thread that takes the TI messages off the
         * TI socket in loop order and hands each to the worker of the pair's peer. */
        const int np_all = c->num_parties;
        ti_pair *pairs = NULL;
        size_t npairs = 0, cap = 0;
        for (size_t i = 0; i <= d; i++)
            for (size_t j = 0; j <= i && j < d; j++) {
                int oi = config_owner(c, i), oj = config_owner(c, j);
                if (oi == oj || (oi != me && oj != me)) continue;
                if (npairs == cap) { cap = cap ? 2 * cap : 1024; pairs = realloc(pairs, cap * sizeof *pairs); }
                ti_pair pr = {oi == me ? oj : oi, oi == me, (uint32_t)(oi == me ? i : j),
                              i < d ? share_A + idx(i, j) : share_b + j};
                pairs[npairs++] = pr;
            }
        /* The TI socket delivers this party's messages in loop order, i.e. in runs of consecutive pairs with
         * the SAME peer (a whole row against one peer's columns).  The queues must hold more than such a
         * run, or the reader blocks on one worker's full queue while the other workers starve. */
        size_t qcap = ((size_t)256 << 20) / (n * 8 + 64);
        if (qcap < 64) qcap = 64;
        if (qcap > 4096) qcap = 4096;
        ti_queue *queues = calloc((size_t)np_all, sizeof *queues);
        ti_worker *workers = calloc((size_t)np_all, sizeof *workers);
        pthread_t *tids = calloc((size_t)np_all, sizeof *tids);
        int started[64] = {0}, failed = 0;
        check(np_all <= 64, "too many parties");
        for (int k = 2; k < np_all; k++) {
            if (k == me) continue;
            size_t cnt = 0;
            for (size_t q = 0; q < npairs; q++) cnt += pairs[q].peer == k;
            if (!cnt) continue;
            ti_queue_init(&queues[k], qcap);
            ti_worker w = {self, p1, n, k, pairs, npairs, &queues[k], 0};
            workers[k] = w;
            if (pthread_create(&tids[k], NULL, ti_worker_main, &workers[k])) { failed = 1; break; }
            started[k] = 1;
        }
        /* reader: this thread */
        double rd_hdr = 0, rd_body = 0, rd_push = 0;
        const int timing_r = getenv("LINREG_TIMING") != NULL;
        for (size_t q = 0; q < npairs && !failed; q++) {
            ti_item it = {0, 0, 0, 0};
            size_t sz = 0;
            double r0 = timing_r ? now_s() : 0, r1, r2;
            if (timing_r) {
                if (net_recv(self, 1, &sz, sizeof sz) || sz > g_pmsg_limit) { failed = 1; break; }
                r1 = now_s();
                if (!(it.raw = malloc(sz ? sz : 1)) || net_recv(self, 1, it.raw, sz)) { failed = 1; free(it.raw); break; }
                r2 = now_s();
                it.len = sz;
                if (ti_queue_push(&queues[pairs[q].peer], it)) { failed = 1; free(it.raw); break; }
                rd_hdr += r1 - r0; rd_body += r2 - r1; rd_push += now_s() - r2;
                continue;
            }
            if (net_recv(self, 1, &sz, sizeof sz) || sz > g_pmsg_limit || !(it.raw = malloc(sz ? sz : 1)) || net_recv(self, 1, it.raw, sz)) {
                fprintf(stderr, "Could not receive message from TI\n"); failed = 1; free(it.raw); break;
            }
            it.len = sz;
            if (ti_queue_push(&queues[pairs[q].peer], it)) { failed = 1; free(it.raw); break; }   /* the worker gave up */
        }
        if (timing_r) fprintf(stderr, "reader: %zu TI messages: waiting for header %.2fs, body %.2fs, queue push %.2fs\n", npairs, rd_hdr, rd_body, rd_push);
        for (int k = 2; k < np_all; k++) if (started[k]) ti_queue_close(&queues[k]);
        for (int k = 2; k < np_all; k++)
            if (started[k]) { pthread_join(tids[k], NULL); failed |= workers[k].failed; ti_queue_destroy(&queues[k]); }
        free(pairs); free(queues); free(workers); free(tids);
        check(!failed, "TI-mode aggregation failed");
        if (getenv("LINREG_TIMING")) fprintf(stderr, "party %d: TI-mode aggregation done after %.2fs\n", c->party, wall_clock() - t_start);
    } else {
        /* OT mode (src/phase1.c:353-450): one Gilboa batch per peer, peers in a global order */
        for (int lo = 2; lo < c->num_parties; lo++)
            for (int hi = lo + 1; hi < c->num_parties; hi++) {
                if (me != lo && me != hi) continue;
                int peer = me == lo ? hi : lo;
                int i_am_sender = ((me % 2 == peer % 2) == (me < peer));          /* phase1.c:392 */
                int pi = i_am_sender ? me : peer, pj = i_am_sender ? peer : me;
                size_t i0 = (size_t)c->index_owned[pi], i1 = pi < last ? (size_t)c->index_owned[pi + 1] : d;
                size_t j0 = (size_t)c->index_owned[pj], j1 = pj < last ? (size_t)c->index_owned[pj + 1] : d;
                size_t npairs = (i1 - i0) * (j1 - j0) + (pj == last ? (i1 - i0) : 0) + (pi == last ? (j1 - j0) : 0);
                /* rows of the sender / receiver per pair, and where the share goes */
                size_t *ri = malloc(npairs * sizeof(size_t)), *rj = malloc(npairs * sizeof(size_t)), q = 0;
                for (size_t i = i0; i < i1; i++) {
                    for (size_t j = j0; j < j1; j++) { ri[q] = i; rj[q++] = j; }
                    if (pj == last) { ri[q] = i; rj[q++] = d; }
                }
                if (pi == last) for (size_t j = j0; j < j1; j++) { ri[q] = d; rj[q++] = j; }
                /* batches of pairs: at most 2^25 OTs (512 MiB of u) each, two in flight on the receiver side */
                size_t per = ((size_t)1 << 25) / (n * (size_t)w1);
                if (per < 1) per = 1;
                if (per > npairs) per = npairs;
                uint64_t *vals = lgc_host_alloc(per * n * 8), *shares = malloc(npairs * 8);
                const uint64_t mmax = (uint64_t)per * n * (uint64_t)w1;
                uint8_t *u = lgc_host_alloc(lgc_ot_u_bytes(mmax));
                uint64_t *yv = lgc_host_alloc(mmax * 8);
                check(vals && u && yv, "%s", lgc_last_error());
                lgc_ot_sender *S = 0;
                lgc_ot_receiver *R = 0;
                if (i_am_sender) {
                    uint8_t delta[16], seeds[128][16];
                    check(!baseot_ext_sender(self, peer + 1, delta, seeds), "base OT failed");
                    LGC(lgc_ot_sender_create(&S, device, delta, seeds));
                } else {
                    uint8_t s0[128][16], s1[128][16];
                    check(!baseot_ext_receiver(self, peer + 1, s0, s1), "base OT failed");
                    LGC(lgc_ot_receiver_create(&R, device, s0, s1));
                }
                double ot_t[4] = {0, 0, 0, 0};
                if (use_ot & 2) {                         /* --ot_ring: u / y through device rings (same node) */
                    check(!ot_pair_ring(self, peer + 1, i_am_sender, S, R, device, Xq, yq, n, d, w1, i_am_sender ? ri : rj, npairs, per, shares),
                          "OT-mode aggregation failed");
                } else if (i_am_sender) {
                    ot_send_ctx sx;
                    memset(&sx, 0, sizeof sx);
                    sx.self = self; sx.peer = peer + 1; sx.n = n; sx.w1 = w1; sx.npairs = npairs; sx.per = per;
                    sx.u[0] = u; sx.y[0] = yv;
                    sx.u[1] = lgc_host_alloc(lgc_ot_u_bytes(mmax)); sx.y[1] = lgc_host_alloc(mmax * 8);
                    check(sx.u[1] && sx.y[1], "%s", lgc_last_error());
                    pthread_mutex_init(&sx.mu, NULL); pthread_cond_init(&sx.cv, NULL);
                    pthread_t tin, tout;
                    check(!pthread_create(&tin, NULL, ot_send_recv_u, &sx), "pthread_create failed");
                    check(!pthread_create(&tout, NULL, ot_send_send_y, &sx), "pthread_create failed");
                    int bad = 0;
                    for (size_t q0 = 0, k = 0; q0 < npairs && !bad; q0 += per, k++) {
                        size_t nb = npairs - q0 < per ? npairs - q0 : per;
                        double t0 = wall_clock();
                        for (q = 0; q < nb; q++) column_of(Xq, yq, n, d, ri[q0 + q], vals + q * n);
                        double t1 = wall_clock();
                        pthread_mutex_lock(&sx.mu);                 /* u of batch k is here, and the y buffer it will fill is free */
                        while ((sx.recvd <= k || k >= sx.sent + 2) && !sx.failed) pthread_cond_wait(&sx.cv, &sx.mu);
                        bad = sx.failed;
                        pthread_mutex_unlock(&sx.mu);
                        if (bad) break;
                        double t2 = wall_clock();
                        if (lgc_ot_gilboa_send(S, vals, nb, n, w1, sx.u[k & 1], sx.y[k & 1], shares + q0)) { fprintf(stderr, "%s\n", lgc_last_error()); bad = 1; }
                        pthread_mutex_lock(&sx.mu); if (bad) sx.failed = 1; else sx.gpu_done = k + 1; pthread_cond_broadcast(&sx.cv); pthread_mutex_unlock(&sx.mu);
                        ot_t[0] += t1 - t0; ot_t[1] += t2 - t1; ot_t[2] += wall_clock() - t2;
                    }
                    pthread_mutex_lock(&sx.mu); if (bad) sx.failed = 1; pthread_cond_broadcast(&sx.cv); pthread_mutex_unlock(&sx.mu);
                    pthread_join(tin, NULL); pthread_join(tout, NULL);
                    bad |= sx.failed;
                    pthread_mutex_destroy(&sx.mu); pthread_cond_destroy(&sx.cv);
                    lgc_host_free(sx.u[1]); lgc_host_free(sx.y[1]);
                    if (getenv("LINREG_TIMING")) fprintf(stderr, "OT sender: columns %.2fs, waiting for u / a free y buffer %.2fs, gpu %.2fs\n", ot_t[0], ot_t[1], ot_t[2]);
                    check(!bad, "OT-mode aggregation failed");
                } else {
                    /* the receiver keeps two batches in flight: a helper thread extends batch k + 1 and sends
                     * its u while this thread waits for the sender's answer to batch k and finishes it */
                    ot_recv_ctx rx = {self, peer + 1, R, Xq, yq, n, d, w1, rj, npairs, per, 0, 0, 0};
                    pthread_mutex_init(&rx.mu, NULL); pthread_cond_init(&rx.cv, NULL);
                    pthread_t th;
                    check(!pthread_create(&th, NULL, ot_recv_starter, &rx), "pthread_create failed");
                    int bad = 0;
                    for (size_t q0 = 0, k = 0; q0 < npairs && !bad; q0 += per, k++) {
                        size_t nb = npairs - q0 < per ? npairs - q0 : per;
                        const uint64_t m = (uint64_t)nb * n * (uint64_t)w1;
                        pthread_mutex_lock(&rx.mu);
                        while (rx.started <= k && !rx.failed) pthread_cond_wait(&rx.cv, &rx.mu);
                        bad = rx.failed;
                        pthread_mutex_unlock(&rx.mu);
                        if (bad) break;
                        if (recv_blob(self, peer + 1, yv, m * 8)) { fprintf(stderr, "OT: could not receive y\n"); bad = 1; }
                        else if (lgc_ot_gilboa_recv_finish(R, yv, shares + q0)) { fprintf(stderr, "%s\n", lgc_last_error()); bad = 1; }
                        pthread_mutex_lock(&rx.mu); rx.finished = k + 1; if (bad) rx.failed = 1; pthread_cond_broadcast(&rx.cv); pthread_mutex_unlock(&rx.mu);
                    }
                    pthread_mutex_lock(&rx.mu); if (bad) rx.failed = 1; pthread_cond_broadcast(&rx.cv); pthread_mutex_unlock(&rx.mu);
                    pthread_join(th, NULL);
                    bad |= rx.failed;
                    pthread_mutex_destroy(&rx.mu); pthread_cond_destroy(&rx.cv);
                    check(!bad, "OT-mode aggregation failed");
                }
                if (S) lgc_ot_sender_destroy(S);
                if (R) lgc_ot_receiver_destroy(R);
                for (q = 0; q < npairs; q++) {
                    if (ri[q] < d && rj[q] < d) share_A[idx(ri[q], rj[q])] += shares[q];
                    else share_b[ri[q] < d ? ri[q] : rj[q]] += shares[q];
                }
                free(ri); free(rj); lgc_host_free(vals); free(shares); lgc_host_free(u); lgc_host_free(yv);
            }
        if (w1 == 32) { for (size_t k = 0; k < T; k++) share_A[k] &= 0xffffffffull; for (size_t k = 0; k < d; k++) share_b[k] &= 0xffffffffull; }
    }
    /* different widths in the two phases: every share is shifted on its own (src/phase1.c:609-638) */
    if (w1 == 64 && w2 == 32) {
        for (size_t k = 0; k < T; k++) share_A[k] = (uint64_t)(uint32_t)(uint64_t)(((int64_t)share_A[k]) >> (precision - precision_p2));
        for (size_t k = 0; k < d; k++) share_b[k] = (uint64_t)(uint32_t)(uint64_t)(((int64_t)share_b[k]) >> (precision - precision_p2));
    }
    *res_A = share_A; *res_b = share_b;
    share_A = share_b = 0;
    rc = 0;
error:
    if (p1) lgc_p1_destroy(p1);
    free(Xq); free(yq); free(share_A); free(share_b); free(va); free(vb); free(tmp); free(tmp2);
    return rc;
}

