/* protocol_int.h -- what the files of the host protocol share among themselves (protocol.c, tables.c, phase1_ti.c,
 * phase1_party.c); not part of what bin/linreg and the benchmark binaries see (protocol.h). */
#ifndef LINREG_PROTOCOL_INT_H
#define LINREG_PROTOCOL_INT_H
#include "protocol.h"

/* protocol.c */
extern size_t g_pmsg_limit;
uint8_t *frame_pmsg(const uint64_t *vec, size_t n, uint64_t value, size_t *len);      /* a length-prefixed proto2 message, malloc'd */
/* readdata.c */
int read_own_columns(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
                            int64_t *Xq, int64_t *yq);
int read_own_columns_threads(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
                             int64_t *Xq, int64_t *yq, int threads);
/* phase1_ti.c */
extern int g_ti_ring;                                    /* --ti_ring (protocol_set_ti_ring) */
void tune_malloc(void);
int run_party_ti_ring(node *self, config *c, lgc_p1 *p1, int device, uint64_t *share_A, uint64_t *share_b);
#endif
