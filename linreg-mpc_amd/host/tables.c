/* tables.c -- phase 2 between the CSP and the Evaluator: the garbled-table stream (socket pipeline with optional lanes, or the
 * device-resident ring and its token protocol) and the program fingerprint check.  Replaces the osend / orecv byte stream inside
 * execYaoProtocol (src/cmd/linreg.c:177).  Split from protocol.c in round 4. */
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
#include "protocol_int.h"

/* ---------------------------------------------------------------- phase 2: table stream */
#define TCHK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, lgc_last_error()); return 1; } } while (0)
/* a trace mark of the host (LINREG_TRACE); in bin/linreg_testhooks also the place where LINREG_DIE_AT=<mark> makes this
 * party kill itself (tests/test_host.py: a party lost at a known point of the protocol) */
/* something the main protocol did a moment ago: the peer watchdog of bin/linreg leaves a party alone while this moves */
static volatile unsigned long g_host_progress;
void host_progress_tick(void) { __atomic_add_fetch(&g_host_progress, 1, __ATOMIC_RELAXED); }
unsigned long host_progress(void) { return __atomic_load_n(&g_host_progress, __ATOMIC_RELAXED); }
void host_trace_mark(const char *what) {
    host_progress_tick();
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
#ifndef LINREG_RING_ASYNC_SHORT
#define LINREG_RING_ASYNC_SHORT 0      /* how a garbling link of fewer than 1000 launches drives them: 0 sync loop, 2 one-stream asynchronous.
                                          Measured at the end of round 5 (scripts/exp/ring_modes3_ab.sh, six runs each on one box): the table
                                          phase of d = 100 CGD-15 0.1489 s (0) / 0.1443 (1) / 0.1457 (2), d = 20 Cholesky 0.0447 / 0.0427 /
                                          0.0436, d = 5 CGD-10 0.0333 / 0.0321 / 0.0319 -- and in bin/linreg the CSP's first blocking copy after
                                          the tables (the decode bits) then takes 9 ms instead of 0.1 (config 3; not in a stand-alone garbler,
                                          scripts/exp/decode_after_async.py), so the Evaluator leaves no earlier: 0 stays */
#endif
enum { kRingAsyncShort = LINREG_RING_ASYNC_SHORT };
static struct { lgc_party *po; ring_hello h; int met; } g_prepared[kMaxPrepared];   /* met: the hello has been exchanged already */
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
    for (int i = 0; i < kMaxPrepared && !ok; i++) if (!g_prepared[i].po) { g_prepared[i].po = po; g_prepared[i].h = h; g_prepared[i].met = 0; ok = 1; }
    pthread_mutex_unlock(&g_prepared_mu);
    return ok ? 0 : 1;
}
/* 0: nothing kept for this party object; 1: a created ring; 2: a ring whose hello both sides have seen */
static int take_prepared(lgc_party *po, ring_hello *h) {
    int found = 0;
    pthread_mutex_lock(&g_prepared_mu);
    for (int i = 0; i < kMaxPrepared && !found; i++)
        if (g_prepared[i].po == po) { *h = g_prepared[i].h; g_prepared[i].po = NULL; found = 1 + g_prepared[i].met; }
    pthread_mutex_unlock(&g_prepared_mu);
    return found;
}
static int keep_met(lgc_party *po, const ring_hello *h) {
    int ok = 0;
    pthread_mutex_lock(&g_prepared_mu);
    for (int i = 0; i < kMaxPrepared && !ok; i++) if (!g_prepared[i].po) { g_prepared[i].po = po; g_prepared[i].h = *h; g_prepared[i].met = 1; ok = 1; }
    pthread_mutex_unlock(&g_prepared_mu);
    return ok ? 0 : 1;
}
static int ring_open_hello(lgc_party *po, const ring_hello *h) {
    if (h->nslots == 0) TCHK(lgc_party_ring_open_bytes(po, h->handle, (size_t)h->slot_bytes));
    else TCHK(lgc_party_ring_open(po, h->handle, (int)h->nslots, (size_t)h->slot_bytes));
    return 0;
}
/* The hello of the party connection ahead of the tables: the garbler names its ring as soon as it exists, the evaluator
 * maps it while the data providers are still in the label OT -- hipIpcOpenMemHandle on a ring of tens of GB takes 20 ms
 * (config 4; 3.6 ms for config 3's), which tables_recv would spend with every input label in and the garbler waiting.
 * tables_send / tables_recv pick the ring up from here (table_link_open). */
int tables_ring_meet(node *self, int peer, lgc_party *po, int sending, int ring_slots) {
    if (ring_slots <= 0) return 0;
    ring_hello h;
    memset(&h, 0, sizeof h);
    if (sending) {
        if (!take_prepared(po, &h) && ring_create(po, ring_slots, &h)) return 1;
        if (send_blob(self, peer, &h, sizeof h)) return 1;
    } else {
        if (recv_blob(self, peer, &h, sizeof h)) return 1;
        if (ring_open_hello(po, &h)) return 1;
    }
    return keep_met(po, &h);
}
int table_link_open(table_link *l, node *self, int peer, int fd, lgc_party *po, int sending, int ring_slots, size_t start) {
    memset(l, 0, sizeof *l);
    l->self = self; l->peer = peer; l->fd = fd; l->po = po; l->start = start; l->end = lgc_party_num_launches(po);
    ring_hello h;
    memset(&h, 0, sizeof h);
    const int kept = take_prepared(po, &h);
    if (kept == 2) {                                                /* tables_ring_meet has been here */
    } else if (sending) {
        if (!kept && ring_create(po, ring_slots, &h)) return 1;
        if (link_io(l, &h, sizeof h, 1)) return 1;
    } else {
        if (link_io(l, &h, sizeof h, 0)) return 1;
        if (ring_open_hello(po, &h)) return 1;
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
/* the second thread of a garbling link (byte ring): waits for launch after launch to complete and sends its token */
typedef struct { table_link *l; size_t lo, hi, begun; int failed; pthread_mutex_t mu; pthread_cond_t cv; } ring_notifier;
static void ring_notifier_post(ring_notifier *n, size_t begun) {
    pthread_mutex_lock(&n->mu); n->begun = begun; pthread_cond_broadcast(&n->cv); pthread_mutex_unlock(&n->mu);
}
static void ring_notifier_fail(ring_notifier *n) {
    pthread_mutex_lock(&n->mu); n->failed = 1; pthread_cond_broadcast(&n->cv); pthread_mutex_unlock(&n->mu);
}
static int ring_notifier_failed(ring_notifier *n) {
    pthread_mutex_lock(&n->mu); int f = n->failed; pthread_mutex_unlock(&n->mu); return f;
}
static void *ring_notify_main(void *arg) {
    ring_notifier *n = arg;
    for (size_t i = n->lo; i < n->hi; i++) {
        pthread_mutex_lock(&n->mu);
        while (n->begun <= i && !n->failed) pthread_cond_wait(&n->cv, &n->mu);
        int stop = n->begun <= i;                                   /* failed before launch i was begun */
        pthread_mutex_unlock(&n->mu);
        if (stop) return NULL;
        if (lgc_party_garble_ring_wait(n->l->po, i) != 0) { fprintf(stderr, "%s\n", lgc_last_error()); ring_notifier_fail(n); return NULL; }
        if (i == 0) host_trace_mark("first table garbled");
        host_progress_tick();
        uint8_t tok = 1;
        if (link_io(n->l, &tok, 1, 1)) { ring_notifier_fail(n); return NULL; }
    }
    return NULL;
}
/* garbler: launches [lo, hi); launch i reuses the slot of launch i - nslots and waits for its ack */
int table_link_send_range(table_link *l, size_t lo, size_t hi) {
    uint8_t tok = 0;
    if (l->nslots == 0) {
        /* byte ring: launch i may overwrite its range once the launch lgc_party_ring_wait_for names has been evaluated
         * (launches before l->start never pass through this link).  The evaluator acknowledges the launches somebody will
         * wait for -- those up to link_last_ack -- and no others; that the evaluator is through with the ring altogether is
         * ONE more byte at the very end (table_link_finish) */
        const int64_t last_ack = link_last_ack(l);
        /* The launches are ENQUEUED by this thread as fast as the ring discipline allows (lgc_party_garble_ring_begin returns at
         * once), and a second thread tells the evaluator about each one when its tables are complete
         * (lgc_party_garble_ring_wait): the garbler's stream never waits for the host, and the table passes of critical-path
         * launches run beside the next launches (rounds 2-4: kernel, device synchronisation, token, next kernel). */
        /* ... where that pays.  The asynchronous path costs a stream (a hardware queue: ~10 ms to create, and ~60 ms more at
         * process exit when several parties leave together), 128 events and a second stash -- 25-30 ms of a
         * short run's wall clock --; it takes the table passes off the chain and ~10 us off every launch.  The table phase
         * itself, evaluator's marks, six runs each on one box (scripts/exp/ring_modes_ab.sh): d = 20 Cholesky (147 launches)
         * 0.0436 s against 0.0448 s with the synchronous loop, d = 100 CGD-15 (355) 0.140 against 0.147, d = 200 Cholesky
         * (1 400) 1.30 against 1.35: 3-5 % of a phase that is a tenth of a small run -- taken from a thousand launches on.
         * LINREG_RING_ASYNC=1 / 0 forces it on / off in bin/linreg_testhooks (the tests run every form on the README example). */
        const char *force = 0;
#ifdef LINREG_TEST_HOOKS
        force = getenv("LINREG_RING_ASYNC");
#endif
        const int use_async = force ? atoi(force) : (l->end - l->start >= 1000 ? 1 : kRingAsyncShort);
        /* 2: the same with the table passes left on the record kernels' stream (lgc_party_garble_ring_streams(po, 1)): no
         * queue to create, and still no host round trip between the garbler's launches */
        if (use_async == 2 && lo == l->start) TCHK(lgc_party_garble_ring_streams(l->po, 1));
        if (!use_async) {                      /* the loop of rounds 2-4: garble, synchronise, tell, next */
            for (size_t i = lo; i < hi; i++) {
                int64_t wf = lgc_party_ring_wait_for(l->po, i);
                size_t need = wf >= (int64_t)l->start ? (size_t)(wf - (int64_t)l->start) + 1 : 0;
                while (l->acked < need) { if (link_io(l, &tok, 1, 0)) return 1; l->acked++; }
                if (i == 0) host_trace_mark("first launch begins");
                TCHK(lgc_party_garble_ring(l->po, i));
                if (i == 0) host_trace_mark("first table garbled");
                host_progress_tick();
                tok = 1;
                if (link_io(l, &tok, 1, 1)) return 1;
            }
            size_t due0 = (last_ack >= (int64_t)l->start) ? (size_t)(last_ack - (int64_t)l->start) + 1 : 0;
            if (due0 > hi - l->start) due0 = hi - l->start;
            while (l->acked < due0) { if (link_io(l, &tok, 1, 0)) return 1; l->acked++; }
            return 0;
        }
        ring_notifier nt;
        memset(&nt, 0, sizeof nt);
        nt.l = l; nt.lo = lo; nt.hi = hi; nt.begun = lo;
        pthread_mutex_init(&nt.mu, NULL);
        pthread_cond_init(&nt.cv, NULL);
        pthread_t th;
        if (pthread_create(&th, NULL, ring_notify_main, &nt)) { fprintf(stderr, "table link: could not start the notifier thread\n"); return 1; }
        int bad = 0;
        for (size_t i = lo; i < hi && !bad; i++) {
            int64_t wf = lgc_party_ring_wait_for(l->po, i);
            size_t need = wf >= (int64_t)l->start ? (size_t)(wf - (int64_t)l->start) + 1 : 0;
            while (l->acked < need && !bad) { if (link_io(l, &tok, 1, 0)) bad = 1; else l->acked++; }
            if (!bad && ring_notifier_failed(&nt)) bad = 1;
            if (!bad && lgc_party_garble_ring_begin(l->po, i) != 0) { fprintf(stderr, "%s\n", lgc_last_error()); bad = 1; }
            if (!bad) ring_notifier_post(&nt, i + 1);
        }
        if (bad) ring_notifier_fail(&nt);
        pthread_join(th, NULL);
        bad |= nt.failed;
        pthread_mutex_destroy(&nt.mu);
        pthread_cond_destroy(&nt.cv);
        if (bad) return 1;
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
        host_progress_tick();
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
        host_progress_tick();
        if (after_launch) after_launch(i, ctx);
        if ((l->nslots == 0 ? (int64_t)i <= link_last_ack(l) : i + l->nslots < l->end) && link_io(l, &tok, 1, 1)) return 1;
    }
    return 0;
}
/* The end of a link.  The ring is the GARBLER's allocation, exported with hipIpcGetMemHandle and mapped by the evaluator:
 * HIP leaves freeing it while the importer still has it open -- and is still reading the last launches -- undefined (it
 * happens to work on this stack because the driver keeps the buffer alive).  So the evaluator says once, with one byte, that
 * every launch of the link has been evaluated (lgc_party_evaluate_ring returns after the launch has run), and the garbler
 * keeps its party object -- the ring -- until then. */
int table_link_finish(table_link *l, int sending) {
    uint8_t tok = 0xE0;
    if (!sending) return link_io(l, &tok, 1, 1);
    /* (every acknowledgement of the ranges has been read by table_link_send_range: this byte is the next on the channel) */
    if (link_io(l, &tok, 1, 0)) return 1;
    if (tok != 0xE0) { fprintf(stderr, "table link: unexpected byte %02x where the evaluator's end-of-ring byte was due\n", tok); return 1; }
    host_progress_tick();
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
                        "--prec_phase2 and --devices must be the same on parties 1 and 2\n");
    return ok ? 0 : 1;
}

int tables_send(node *self, int peer, lgc_party *po, int ring_slots, size_t chunk) {
    const size_t nl = lgc_party_num_launches(po);
    if (ring_slots > 0) {
        table_link l;
        if (table_link_open(&l, self, peer, -1, po, 1, ring_slots, 0)) return 1;
        return table_link_send_range(&l, 0, nl) || table_link_finish(&l, 1);
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
        host_progress_tick();
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
        return table_link_recv_range(&l, 0, nl, after_launch, ctx) || table_link_finish(&l, 0);
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
        host_progress_tick();
        if (after_launch) after_launch(i, ctx);
    }
    table_pipe_stop(&tp, th);
    int rc = table_pipe_failed(&tp);
    table_pipe_free(&tp);
    return rc;
}
