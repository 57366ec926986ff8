/* net.h -- TCP mesh between parties, by party number (counterpart of the reference's
 * src/node.c:11-70 and the osend/orecv calls of Obliv-C's ProtocolDesc).  Blocking sockets;
 * every party connects to all lower-numbered parties and accepts all higher-numbered ones. */
#ifndef LINREG_NET_H
#define LINREG_NET_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int party;          /* 1-based */
    int num_parties;
    int *fd;            /* fd[q-1] = socket to party q, -1 for self */
    uint64_t *sent;     /* bytes sent per peer (PROFILE_NETWORK-style accounting) */
    uint64_t *wait_ns;  /* time spent waiting for a peer data provider's message, per peer (src/phase1.c:177-183) */
    uint64_t *nsend;    /* send calls per peer */
    /* Obliv-C's -DPROFILE_NETWORK counts FLUSHES of its buffered transport, not writes: one per explicit
     * flush (the orecv(pd,0,NULL,0) idiom: after every protobuf message, src/phase1.c:141; after the party
     * announcement, src/node.c:37; at the end of a barrier, src/cmd/secure_multiplication.c:31), one whenever
     * a read finds unflushed output pending, and one when the connection is cleaned up.  The sockets here
     * are unbuffered, so the same events are counted where the reference's code has them. */
    uint64_t *pending;  /* bytes written to a peer since its last flush event */
    uint64_t *nflush;   /* flush events per peer (net_flush_count adds the one of the final cleanup) */
} node;

int node_new(node **out, int party, int num_parties, char **endpoints);
void node_destroy(node **n);
int net_send(node *n, int to_party, const void *buf, size_t len);
int net_recv(node *n, int from_party, void *buf, size_t len);
void net_flush(node *n, int to_party);               /* explicit flush point */
int net_send_flush(node *n, int to_party, const void *buf, size_t len);   /* send followed by an explicit flush */
uint64_t net_flush_count(const node *n, int party);  /* as "Total flush done" would report it at cleanup */
/* Extra TCP connections to one peer (bulk data striped over several streams: one stream is one core's worth of
 * copying on either side).  The side that calls net_lanes_offer listens on an ephemeral port, tells the peer over the
 * main connection and accepts k connections; the peer calls net_lanes_accept_offer and connects to the address the
 * main connection comes from.  fds[i] is lane i on both sides.  Returns 0 on success. */
int net_lanes_offer(node *n, int peer, int k, int *fds);
int net_lanes_accept_offer(node *n, int peer, int max_k, int *k, int *fds);
int net_io_all(int fd, void *buf, size_t len, int wr);   /* full-length send (wr = 1) / recv on a raw descriptor */
int net_barrier(node *n);    /* chain barrier of src/cmd/linreg.c:19-41 */
double wall_clock(void);
#endif
