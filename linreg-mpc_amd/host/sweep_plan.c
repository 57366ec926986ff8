/* sweep_plan.c -- which circuits of a --lambdas sweep go to which entry of --devices: pure host logic (no GPU), shared by
 * bin/linreg and the CPU tests (libhosttest.so).  The plan depends on the NUMBER of entries only, never on the indices: the
 * blocks of --devices=0,1 are the blocks of --devices=0,0 (tests/test_host.py), so what a one-GPU box rehearses is what two
 * GPUs run.  Reference: one execYaoProtocol per lambda (src/cmd/linreg.c:177 inside the callers' lambda loops). */
#include <stdlib.h>

#include "protocol.h"

/* contiguous blocks of (nearly) equal size: the first n % K blocks hold one circuit more */
void sweep_block_range(size_t n, size_t K, size_t k, size_t *lo, size_t *hi) {
    size_t base = n / K, extra = n % K;
    *lo = k * base + (k < extra ? k : extra);
    *hi = *lo + base + (k < extra ? 1 : 0);
}

/* "g0,g1,..." -> devices[]; returns the number of entries, or -1 (malformed, negative, more than `max`) */
int sweep_parse_devices(const char *text, int *devices, int max) {
    int n = 0;
    const char *q = text;
    if (!q || !*q) return -1;
    while (*q) {
        char *e;
        long v = strtol(q, &e, 10);
        if (e == q || (*e != ',' && *e) || v < 0 || v > 1 << 20 || n >= max) return -1;
        devices[n++] = (int)v;
        q = *e ? e + 1 : e;
        if (*e == ',' && !*q) return -1;               /* trailing comma */
    }
    return n;
}

/* the whole plan: no empty blocks (entries beyond the number of circuits are dropped); returns the number of blocks */
int sweep_plan(size_t n_lambdas, const int *devices, int n_devices, sweep_block *plan) {
    if (n_devices <= 0 || n_lambdas == 0) return 0;
    if ((size_t)n_devices > n_lambdas) n_devices = (int)n_lambdas;
    for (int k = 0; k < n_devices; k++) {
        plan[k].device = devices[k];
        sweep_block_range(n_lambdas, (size_t)n_devices, (size_t)k, &plan[k].lo, &plan[k].hi);
    }
    return n_devices;
}
