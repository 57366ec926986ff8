/* asan_host.c -- the host's two parsers of untrusted bytes (the TI-mode wire message, host/pmsg.c, and the input-file
 * header, host/config.c) under AddressSanitizer + UBSan: round trips, then mutated and truncated inputs.  A decoder
 * may reject what it is given; it must not read or write outside its buffers (each input is a heap block of exactly
 * its length).  tests/test_host.py compiles and runs it.
 *
 *   gcc -O1 -g -std=gnu11 -fsanitize=address,undefined -fno-sanitize-recover=all -I linreg-mpc_amd/host \
 *       tests/tools/asan_host.c linreg-mpc_amd/host/pmsg.c linreg-mpc_amd/host/config.c -o asan_host */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "config.h"
#include "pmsg.h"

static uint64_t s = 0x2545f4914f6cdd1dull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static int fail(const char *what) { printf("FAIL: %s\n", what); return 1; }

static int pmsg_cases(void) {
    int bad = 0;
    for (int it = 0; it < 3000; it++) {
        const size_t n = rnd() % 70;
        uint64_t *v = malloc(sizeof(uint64_t) * (n ? n : 1));
        for (size_t i = 0; i < n; i++) v[i] = rnd() >> (rnd() % 64);          /* every varint length */
        const uint64_t value = rnd() >> (rnd() % 64);
        const size_t len = pmsg_packed_size(v, n, value);
        uint8_t *buf = malloc(len ? len : 1);
        if (pmsg_pack(v, n, value, buf) != len) bad |= fail("pmsg_pack length");
        uint64_t *out = 0, val = 0;
        size_t m = 0;
        if (pmsg_unpack(buf, len, &out, &m, &val) != 0 || m != n || val != value || (n && memcmp(out, v, n * 8))) bad |= fail("pmsg round trip");
        free(out);
        uint64_t *into = malloc(sizeof(uint64_t) * (n ? n : 1));
        if (pmsg_unpack_into(buf, len, into, n, &m, &val) != 0 || m != n || (n && memcmp(into, v, n * 8))) bad |= fail("pmsg_unpack_into round trip");
        if (n > 0 && pmsg_unpack_into(buf, len, into, n - 1, &m, &val) == 0) bad |= fail("pmsg_unpack_into accepted more elements than cap");
        free(into);
        /* truncations: every prefix, each in a block of its own length */
        for (size_t cut = 0; cut < len; cut += 1 + cut / 16) {
            uint8_t *t = malloc(cut ? cut : 1);
            memcpy(t, buf, cut);
            out = 0;
            if (pmsg_unpack(t, cut, &out, &m, &val) == 0) free(out);
            uint64_t *small = malloc(sizeof(uint64_t) * (n ? n : 1));
            (void)pmsg_unpack_into(t, cut, small, n, &m, &val);
            free(small);
            free(t);
        }
        /* mutations: flipped bytes, forged lengths */
        for (int k = 0; k < 8 && len; k++) {
            uint8_t *t = malloc(len);
            memcpy(t, buf, len);
            const int flips = 1 + (int)(rnd() % 3);
            for (int f = 0; f < flips; f++) t[rnd() % len] = (uint8_t)rnd();
            out = 0;
            if (pmsg_unpack(t, len, &out, &m, &val) == 0) free(out);
            uint64_t *small = malloc(sizeof(uint64_t) * (n ? n : 1));
            (void)pmsg_unpack_into(t, len, small, n, &m, &val);
            free(small);
            free(t);
        }
        free(buf);
        free(v);
    }
    /* pure noise */
    for (int it = 0; it < 20000; it++) {
        const size_t len = rnd() % 48;
        uint8_t *t = malloc(len ? len : 1);
        for (size_t i = 0; i < len; i++) t[i] = (uint8_t)rnd();
        uint64_t *out = 0, val;
        size_t m;
        if (pmsg_unpack(t, len, &out, &m, &val) == 0) free(out);
        uint64_t small[4];
        (void)pmsg_unpack_into(t, len, small, 4, &m, &val);
        free(t);
    }
    return bad;
}

static int write_file(const char *path, const char *text, size_t len) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(text, 1, len, f);
    fclose(f);
    return 0;
}

static int config_cases(const char *dir) {
    int bad = 0;
    char path[512];
    snprintf(path, sizeof path, "%s/asan_host_cfg.in", dir);
    const char *good = "4 3 2\nlocalhost:1234\nlocalhost:1235\nlocalhost:1236 0\nlocalhost:1237 2\n1 2 3\n4 5 6\n7 8 9\n10 11 12\n1 2 3 4\n";
    const size_t glen = strlen(good);
    if (write_file(path, good, glen)) return fail("cannot write a temporary file");
    config *c = 0;
    if (config_new(&c, path) != 0 || !c) return fail("config_new rejected a well-formed header");
    if (c->n != 4 || c->d != 3 || c->num_parties != 4) bad |= fail("config fields");
    if (config_owner(c, 0) != 2 || config_owner(c, 1) != 2 || config_owner(c, 2) != 3 || config_owner(c, 3) != 3) bad |= fail("config_owner");
    config_destroy(&c);
    /* every truncation of the header, then mutated headers: errors are fine, stray accesses are not */
    for (size_t cut = 0; cut < glen; cut++) {
        write_file(path, good, cut);
        c = 0;
        if (config_new(&c, path) == 0 && c) {
            for (size_t r = 0; r <= c->d && r < 64; r++) (void)config_owner(c, r);
            config_destroy(&c);
        }
    }
    for (int it = 0; it < 1500; it++) {
        char *t = malloc(glen + 1);
        memcpy(t, good, glen + 1);
        const int flips = 1 + (int)(rnd() % 4);
        for (int f = 0; f < flips; f++) {
            const char alphabet[] = "0123456789 -\n:x";
            t[rnd() % 60] = alphabet[rnd() % (sizeof alphabet - 1)];
        }
        write_file(path, t, glen);
        c = 0;
        if (config_new(&c, path) == 0 && c) {
            for (size_t r = 0; r <= c->d && r < 64; r++) (void)config_owner(c, r);
            config_destroy(&c);
        }
        free(t);
    }
    unlink(path);
    return bad;
}

int main(int argc, char **argv) {
    int bad = pmsg_cases();
    bad |= config_cases(argc > 1 ? argv[1] : "/tmp");
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
