#include "net.h"

#include <arpa/inet.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

double wall_clock(void) {
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static int io_all(int fd, void *buf, size_t len, int wr) {
    char *p = buf;
    while (len) {
        ssize_t k = wr ? send(fd, p, len, MSG_NOSIGNAL) : recv(fd, p, len, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return -1;
        p += k; len -= (size_t)k;
    }
    return 0;
}

int net_io_all(int fd, void *buf, size_t len, int wr) { return io_all(fd, buf, len, wr); }

int net_send(node *n, int to, const void *buf, size_t len) {
    if (to < 1 || to > n->num_parties || n->fd[to - 1] < 0) return -1;
    n->sent[to - 1] += len;
    n->nsend[to - 1]++;
    __atomic_fetch_add(&n->pending[to - 1], len, __ATOMIC_RELAXED);
    return io_all(n->fd[to - 1], (void *)buf, len, 1);
}
int net_recv(node *n, int from, void *buf, size_t len) {
    if (from < 1 || from > n->num_parties || n->fd[from - 1] < 0) return -1;
    /* a buffered transport flushes its pending output before it blocks in a read */
    if (__atomic_exchange_n(&n->pending[from - 1], 0, __ATOMIC_RELAXED)) __atomic_fetch_add(&n->nflush[from - 1], 1, __ATOMIC_RELAXED);
    return io_all(n->fd[from - 1], buf, len, 0);
}
/* send + explicit flush as one accounting event (a concurrent reader on the same connection must not see the
 * message as pending output and count a second, implicit flush) */
int net_send_flush(node *n, int to, const void *buf, size_t len) {
    if (to < 1 || to > n->num_parties || n->fd[to - 1] < 0) return -1;
    n->sent[to - 1] += len;
    n->nsend[to - 1]++;
    __atomic_store_n(&n->pending[to - 1], 0, __ATOMIC_RELAXED);
    __atomic_fetch_add(&n->nflush[to - 1], 1, __ATOMIC_RELAXED);
    return io_all(n->fd[to - 1], (void *)buf, len, 1);
}
void net_flush(node *n, int to) {
    if (to < 1 || to > n->num_parties || n->fd[to - 1] < 0) return;
    __atomic_store_n(&n->pending[to - 1], 0, __ATOMIC_RELAXED);
    __atomic_fetch_add(&n->nflush[to - 1], 1, __ATOMIC_RELAXED);
}
uint64_t net_flush_count(const node *n, int party) {
    if (party < 1 || party > n->num_parties || n->fd[party - 1] < 0) return 0;
    return n->nflush[party - 1] + 1;                   /* + the flush of cleanupProtocol */
}

static int split_endpoint(const char *ep, char *host, size_t hl, char *port, size_t pl) {
    const char *c = strrchr(ep, ':');
    if (!c || (size_t)(c - ep) >= hl || strlen(c + 1) >= pl) return -1;
    memcpy(host, ep, (size_t)(c - ep)); host[c - ep] = 0;
    strcpy(port, c + 1);
    return 0;
}

/* large socket buffers: a TI message is ~0.5 MB at n = 5*10^4; with the default ~200 KB buffers a
 * sender sleeps several times per message (the kernel caps the request at net.core.[rw]mem_max) */
static void tune_socket(int s) {
    int one = 1, big = 8 << 20;
    setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    setsockopt(s, SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
    setsockopt(s, SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
    if (getenv("LINREG_TIMING")) {
        int sb = 0, rb = 0; socklen_t l = sizeof sb;
        getsockopt(s, SOL_SOCKET, SO_SNDBUF, &sb, &l); l = sizeof rb;
        getsockopt(s, SOL_SOCKET, SO_RCVBUF, &rb, &l);
        fprintf(stderr, "socket buffers: snd %d rcv %d\n", sb, rb);
    }
}
/* How long a party waits for its peers to come up (connect retries, accept): LINREG_CONNECT_TIMEOUT seconds, default 300;
 * 0 = for ever, as util_loop_connect (src/util.c:26-38) does.  A party whose peer never appears exits non-zero instead of
 * spinning: with the others gone (check() -> exit 1 everywhere, src/check_error.h) nobody would ever end it. */
static double connect_deadline_s(void) {
    const char *e = getenv("LINREG_CONNECT_TIMEOUT");
    return e && *e ? atof(e) : 300.0;
}
static double mono_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* a connect() to a local port nobody listens on yet can be given that very port as its source and then completes as a TCP
 * simultaneous open WITH ITSELF (ports inside the ephemeral range; once in ~10^4 attempts): not a peer */
static int connected_to_itself(int s) {
    struct sockaddr_in a, b;
    socklen_t la = sizeof a, lb = sizeof b;
    if (getsockname(s, (struct sockaddr *)&a, &la) || getpeername(s, (struct sockaddr *)&b, &lb)) return 0;
    return a.sin_family == AF_INET && b.sin_family == AF_INET && a.sin_port == b.sin_port && a.sin_addr.s_addr == b.sin_addr.s_addr;
}
static int connect_retry(const char *host, const char *port) {
    /* retry like util_loop_connect (src/util.c:26-38), which sleeps 200 ms between attempts: here the pause starts at 2 ms
     * and doubles up to those 200 ms -- parties started together find each other within milliseconds (a peer that was not
     * listening yet used to cost 0.2 s of a 0.4 s run) */
    long pause_ns = 2000000;
    const double limit = connect_deadline_s(), t0 = mono_s();
    for (;;) {
        struct addrinfo hints, *res = 0;
        memset(&hints, 0, sizeof hints);
        hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(host, port, &hints, &res) == 0) {
            int s = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
            if (s >= 0 && connect(s, res->ai_addr, res->ai_addrlen) == 0 && !connected_to_itself(s)) {
                tune_socket(s);
                freeaddrinfo(res);
                return s;
            }
            if (s >= 0) close(s);
            freeaddrinfo(res);
        }
        if (limit > 0 && mono_s() - t0 > limit) {
            fprintf(stderr, "Could not connect to %s:%s within %.0f s (LINREG_CONNECT_TIMEOUT)\n", host, port, limit);
            return -1;
        }
        struct timespec ts = {0, pause_ns};
        nanosleep(&ts, 0);
        if (pause_ns < 200000000) pause_ns *= 2;
        if (pause_ns > 200000000) pause_ns = 200000000;
    }
}

int node_new(node **out, int party, int num_parties, char **endpoints) {
    node *n = calloc(1, sizeof *n);
    if (!n) return 1;
    n->party = party; n->num_parties = num_parties;
    n->fd = malloc(sizeof(int) * (size_t)num_parties);
    n->sent = calloc((size_t)num_parties, sizeof(uint64_t));
    n->nsend = calloc((size_t)num_parties, sizeof(uint64_t));
    n->wait_ns = calloc((size_t)num_parties, sizeof(uint64_t));
    n->pending = calloc((size_t)num_parties, sizeof(uint64_t));
    n->nflush = calloc((size_t)num_parties, sizeof(uint64_t));
    for (int i = 0; i < num_parties; i++) n->fd[i] = -1;
    char host[256], port[32];
    for (int q = 1; q < party; q++) {                       /* lower-numbered peers listen */
        if (split_endpoint(endpoints[q - 1], host, sizeof host, port, sizeof port)) goto fail;
        int s = connect_retry(host, port);
        if (s < 0) goto fail;
        n->fd[q - 1] = s;
        int32_t me = party;
        if (io_all(s, &me, sizeof me, 1)) goto fail;      /* announce ourselves (node.c:35-37) ... */
        n->sent[q - 1] += sizeof me; n->nsend[q - 1]++;
        net_flush(n, q);                                  /* ... and flush */
    }
    if (party < num_parties) {
        if (split_endpoint(endpoints[party - 1], host, sizeof host, port, sizeof port)) goto fail;
        int ls = socket(AF_INET, SOCK_STREAM, 0), one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        struct sockaddr_in sa;
        memset(&sa, 0, sizeof sa);
        sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)atoi(port)); sa.sin_addr.s_addr = INADDR_ANY;
        if (bind(ls, (struct sockaddr *)&sa, sizeof sa) < 0 || listen(ls, SOMAXCONN) < 0) {
            fprintf(stderr, "Could not create listen socket on port %s: %s\n", port, strerror(errno));
            close(ls);
            goto fail;
        }
        const double limit = connect_deadline_s(), t0 = mono_s();
        for (int k = party; k < num_parties; k++) {
            if (limit > 0) {                              /* accept() honours SO_RCVTIMEO */
                double left = limit - (mono_s() - t0);
                struct timeval tv = {left > 1 ? (time_t)left : 1, 0};
                setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            }
            int s = accept(ls, 0, 0);
            if (s < 0) {
                fprintf(stderr, "Party %d: %d of its peers did not connect within %.0f s (LINREG_CONNECT_TIMEOUT): %s\n", party,
                        num_parties - k, limit, strerror(errno));
                close(ls);
                goto fail;
            }
            {   /* the timeout was for accept() only (an accepted socket does not inherit it on Linux, but say so) */
                struct timeval none = {0, 0};
                setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
            }
            tune_socket(s);
            int32_t other = 0;
            if (io_all(s, &other, sizeof other, 0) || other <= party || other > num_parties || n->fd[other - 1] >= 0) {
                fprintf(stderr, "Party %d received invalid party number %d from remote\n", party, other);
                close(s); close(ls);
                goto fail;
            }
            n->fd[other - 1] = s;
        }
        close(ls);
    }
    *out = n;
    return 0;
fail:
    node_destroy(&n);
    return 1;
}

void node_destroy(node **nn) {
    if (!nn || !*nn) return;
    node *n = *nn;
    if (n->fd) for (int i = 0; i < n->num_parties; i++) if (n->fd[i] >= 0) close(n->fd[i]);
    free(n->fd); free(n->sent); free(n->nsend); free(n->wait_ns); free(n->pending); free(n->nflush); free(n);
    *nn = 0;
}

/* Extra connections to a peer (table links of bin/linreg --devices, --table_lanes).  The offering side listens on the
 * LOCAL ADDRESS OF THE PARTY CONNECTION only, tells the peer port and a random 16-byte cookie over that connection,
 * and accepts a lane only from the peer's address and only with the cookie; accept and the cookie read time out. */
typedef struct { uint32_t k, port; uint8_t cookie[16]; } lane_offer;
typedef struct { uint32_t lane; uint8_t cookie[16]; } lane_hello;
enum { kLaneTimeoutSec = 30, kHelloTimeoutSec = 2 };
static void set_timeouts(int s, int sec) {
    struct timeval tv = {sec, 0};
    (void)setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    (void)setsockopt(s, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}
int net_lanes_offer(node *n, int peer, int k, int *fds) {
    if (peer < 1 || peer > n->num_parties || n->fd[peer - 1] < 0 || k < 0) return -1;
    lane_offer o;
    memset(&o, 0, sizeof o);
    if (k == 0) return net_send(n, peer, &o, sizeof o) ? -1 : 0;      /* "no lanes": the peer still expects the offer */
    struct sockaddr_in sa, pa;
    socklen_t sl = sizeof sa, pl = sizeof pa;
    if (getsockname(n->fd[peer - 1], (struct sockaddr *)&sa, &sl) < 0 || sa.sin_family != AF_INET) return -1;
    if (getpeername(n->fd[peer - 1], (struct sockaddr *)&pa, &pl) < 0 || pa.sin_family != AF_INET) return -1;
    int ls = socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return -1;
    sa.sin_port = 0;                                     /* the interface the peer already reaches us on, any free port */
    sl = sizeof sa;
    if (bind(ls, (struct sockaddr *)&sa, sizeof sa) < 0 || listen(ls, k) < 0 || getsockname(ls, (struct sockaddr *)&sa, &sl) < 0) {
        close(ls);
        return -1;
    }
    o.k = (uint32_t)k; o.port = (uint32_t)ntohs(sa.sin_port);
    FILE *ur = fopen("/dev/urandom", "rb");
    if (!ur || fread(o.cookie, 1, sizeof o.cookie, ur) != sizeof o.cookie) { if (ur) fclose(ur); close(ls); return -1; }
    fclose(ur);
    if (net_send(n, peer, &o, sizeof o)) { close(ls); return -1; }
    net_flush(n, peer);
    for (int i = 0; i < k; i++) fds[i] = -1;
    /* ONE deadline for the whole handshake, and a short one per connection for its hello: a host that can reach the port
     * from the peer's address (same node, NAT) and sends nothing costs kHelloTimeoutSec of it, not 30 s per connection */
    struct timespec t0, tn;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int got = 0, strays = 0;
    while (got < k) {
        clock_gettime(CLOCK_MONOTONIC, &tn);
        long left = kLaneTimeoutSec - (long)(tn.tv_sec - t0.tv_sec);
        if (left <= 0) break;
        set_timeouts(ls, (int)left);                     /* accept() honours SO_RCVTIMEO */
        struct sockaddr_in ca;
        socklen_t cl = sizeof ca;
        int s = accept(ls, (struct sockaddr *)&ca, &cl);
        if (s < 0) break;                                /* timed out or failed */
        lane_hello h;
        set_timeouts(s, left < kHelloTimeoutSec ? (int)left : kHelloTimeoutSec);
        unsigned char diff = 0;
        int ok = ca.sin_family == AF_INET && ca.sin_addr.s_addr == pa.sin_addr.s_addr && !io_all(s, &h, sizeof h, 0);
        if (ok) { for (size_t b = 0; b < sizeof o.cookie; b++) diff |= (unsigned char)(h.cookie[b] ^ o.cookie[b]); }
        if (!ok || diff || h.lane >= (uint32_t)k || fds[h.lane] >= 0) {       /* not our peer: drop it and keep listening */
            close(s);
            if (++strays > 64) break;
            continue;
        }
        set_timeouts(s, 0);
        tune_socket(s);
        fds[h.lane] = s;
        got++;
    }
    close(ls);
    if (got < k) {
        for (int j = 0; j < k; j++) if (fds[j] >= 0) { close(fds[j]); fds[j] = -1; }
        return -1;
    }
    return 0;
}
int net_lanes_accept_offer(node *n, int peer, int max_k, int *k, int *fds) {
    if (peer < 1 || peer > n->num_parties || n->fd[peer - 1] < 0) return -1;
    lane_offer o;
    if (net_recv(n, peer, &o, sizeof o)) return -1;
    if (o.k == 0) { *k = 0; return 0; }
    if ((int)o.k > max_k || o.port == 0 || o.port > 65535) return -1;
    struct sockaddr_in sa;
    socklen_t sl = sizeof sa;
    if (getpeername(n->fd[peer - 1], (struct sockaddr *)&sa, &sl) < 0 || sa.sin_family != AF_INET) return -1;
    sa.sin_port = htons((uint16_t)o.port);
    for (uint32_t i = 0; i < o.k; i++) {
        lane_hello h;
        h.lane = i;
        memcpy(h.cookie, o.cookie, sizeof h.cookie);
        int s = socket(AF_INET, SOCK_STREAM, 0);
        if (s < 0 || connect(s, (struct sockaddr *)&sa, sizeof sa) < 0 || io_all(s, &h, sizeof h, 1)) {
            if (s >= 0) close(s);
            for (uint32_t j = 0; j < i; j++) close(fds[j]);
            return -1;
        }
        tune_socket(s);
        fds[i] = s;
    }
    *k = (int)o.k;
    return 0;
}

int net_barrier(node *n) {
    int32_t flag = 42;
    if (n->party != 1 && net_recv(n, n->party - 1, &flag, sizeof flag)) return -1;
    if (n->party != n->num_parties) {
        if (net_send(n, n->party + 1, &flag, sizeof flag)) return -1;
        if (net_recv(n, n->party + 1, &flag, sizeof flag)) return -1;
    }
    if (n->party != 1) {
        if (net_send(n, n->party - 1, &flag, sizeof flag)) return -1;
        net_flush(n, n->party - 1);
    }
    return 0;
}
