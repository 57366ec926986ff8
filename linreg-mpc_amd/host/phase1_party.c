/* phase1_party.c -- phase 1, data-provider side (run_party, src/phase1.c:453-656): TI mode over sockets with per-peer workers,
 * OT mode (Gilboa products over the IKNP extension), the 64 -> 32 bit conversion of the shares.  Split from protocol.c in round 4. */
#define _GNU_SOURCE
#include <errno.h>
#include <malloc.h>
#include <math.h>
#include <openssl/rand.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <sys/mman.h>
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
#include "protocol_int.h"

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
        if (cnt && (as_b == cnt || as_b == 0)) {
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
    /* 200 MB at config 4, touched for the first time by the parser's threads while the HIP runtime comes up on another
     * thread: huge pages mean hundreds of page faults instead of 50 000 fighting that thread for the address-space lock */
    if (Xq && n * d * 8 >= ((size_t)8 << 20)) {
        const uintptr_t a = ((uintptr_t)Xq + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)Xq + n * d * 8) & ~(uintptr_t)4095;
        (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);
    }
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
                /* batches of pairs: at most 2^24 OTs (256 MiB of u) each, two in flight on the receiver side.  Measured on config 3
                 * (1.6e9 OTs between two provider processes on one GPU, scripts/exp/ot_batch_ab.sh): phase 1 through at 0.46 s with
                 * 2^24, 0.58 s with 2^25 (rounds 2-3), 0.8 / 1.2 / 2.2 s with 2^26..28 -- the session's buffers grow with the batch
                 * and a fresh device or page-locked allocation costs more than the per-batch tokens do */
                const int ot_log = 24;
                size_t per = ((size_t)1 << ot_log) / (n * (size_t)w1);
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
