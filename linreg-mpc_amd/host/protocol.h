/* protocol.h -- shared host-side protocol drivers (see protocol.c) */
#ifndef LINREG_PROTOCOL_H
#define LINREG_PROTOCOL_H
#include <stdint.h>
#include <stdio.h>
#include "config.h"
#include "net.h"
#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"

#define check(cond, ...)                              \
    do {                                              \
        if (!(cond)) {                                \
            fprintf(stderr, __VA_ARGS__);             \
            fprintf(stderr, "\n");                    \
            goto error;                               \
        }                                             \
    } while (0)
#define LGC(x) check((x) == 0, "%s: %s", #x, lgc_last_error())

size_t idx(size_t i, size_t j);
int64_t double_to_fixed(double d, int p, int w);
double fixed_to_double(int64_t f, int p);
int read_values(FILE *f, size_t count, int precision, double normalizer, int w2, int64_t *out);
int send_pmsg(node *self, int to, const uint64_t *vec, size_t n, uint64_t value);
int recv_pmsg(node *self, int from, uint64_t **vec, size_t *n, uint64_t *value);
int send_blob(node *self, int to, const void *buf, uint64_t len);
int recv_blob(node *self, int from, void *buf, uint64_t len);
/* CSP (sending = 1) and Evaluator (0) compare the fingerprints of their programs (lgc_party_program_fingerprint) before the
 * table stream starts; both sides report a mismatch.  0 = the programs agree */
int programs_agree(node *self, int peer, lgc_party *po, int sending);
void pmsg_set_limit(size_t n_elements);
void protocol_set_ti_ring(int on);        /* TI mode with all parties on one node: vectors through device rings */   /* bound on the length prefix recv_pmsg accepts */

/* The garbled-table stream of phase 2 (garbler -> evaluator; the reference's Yao runtime does this
 * with osend/orecv inside execYaoProtocol, linreg.c:177).  ring_slots == 0: the table bytes of each
 * launch travel over the socket.  ring_slots > 0: both processes are on one node; the tables stay
 * in a device-resident ring shared through hipIpc and only one-byte ready/ack tokens travel.
 * after_launch (may be NULL) is called once launch i has been evaluated. */
void protocol_set_table_lanes(int k);   /* --table_lanes=K: socket-mode table stream striped over K extra connections */
int tables_send(node *self, int peer, lgc_party *po, int ring_slots, size_t chunk);
int tables_recv(node *self, int peer, lgc_party *po, int ring_slots, size_t chunk,
                void (*after_launch)(size_t launch, void *ctx), void *ctx);
/* ring mode in pieces, for several blocks of a sweep side by side (bin/linreg --devices): see protocol.c */
/* ring_slots: 0 = no ring (tables through the socket), 1..64 = a ring of that many slots of the largest launch,
 * TABLE_RING_BYTES = the byte ring (largest launch + slack: lgc_party_ring_create_bytes), what plain --table_ring selects */
/* sweep_plan.c: the blocks of a --lambdas sweep over the entries of --devices (pure host logic) */
typedef struct { int device; size_t lo, hi; } sweep_block;
void sweep_block_range(size_t n, size_t K, size_t k, size_t *lo, size_t *hi);
int sweep_parse_devices(const char *text, int *devices, int max);
int sweep_plan(size_t n_lambdas, const int *devices, int n_devices, sweep_block *plan);
void host_trace_mark(const char *what);       /* lgc_trace_mark, and the LINREG_DIE_AT hook of bin/linreg_testhooks */
enum { TABLE_RING_BYTES = 65 };
typedef struct { node *self; int peer, fd; lgc_party *po; size_t start, end, nslots, acked; int64_t last_ack; int last_ack_known; } table_link;   /* nslots 0: byte ring */
int tables_ring_prepare(lgc_party *po, int ring_slots);   /* garbler, optional: create the ring before tables_send / table_link_open need it */
/* both, optional, over the party connection: the ring's hello ahead of tables_send / tables_recv (the evaluator maps the ring here) */
int tables_ring_meet(node *self, int peer, lgc_party *po, int sending, int ring_slots);
int table_link_open(table_link *l, node *self, int peer, int fd, lgc_party *po, int sending, int ring_slots, size_t start);
int table_link_send_range(table_link *l, size_t lo, size_t hi);
int table_link_recv_range(table_link *l, size_t lo, size_t hi, void (*after_launch)(size_t launch, void *ctx), void *ctx);
int table_link_finish(table_link *l, int sending);      /* after the last range: the evaluator releases the ring with one byte, the garbler waits for it */
void host_progress_tick(void);                          /* the main protocol moved (every launch, every trace mark) ... */
unsigned long host_progress(void);                      /* ... read by the peer watchdog of bin/linreg */
int run_trusted_initializer(node *self, config *c, int w1, int device);
int run_party(node *self, config *c, int precision, int precision_p2, int w1, int w2, int use_ot, int device,
              uint64_t **res_A, uint64_t **res_b);
#endif
