/* phase1_ti.c -- phase 1, trusted-initializer side (run_trusted_initializer, src/phase1.c:241-339) and the one-node ring form of
 * the TI protocol for both the initializer and the data providers (--ti_ring).  Split from protocol.c in round 4. */
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
void tune_malloc(void) {
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
int g_ti_ring = 0;
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
 * which two thirds were those fixed costs. */
static size_t ti_ring_batch(size_t n) {
    const size_t slot_mb = 256;
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

int run_party_ti_ring(node *self, config *c, lgc_p1 *p1, int device, uint64_t *share_A, uint64_t *share_b) {
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
