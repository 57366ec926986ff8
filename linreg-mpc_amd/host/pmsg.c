#include "pmsg.h"

#include <stdlib.h>

static size_t varint_size(uint64_t v) { size_t s = 1; while (v >= 0x80) { v >>= 7; s++; } return s; }
static size_t put_varint(uint64_t v, uint8_t *o) {
    size_t s = 0;
    while (v >= 0x80) { o[s++] = (uint8_t)(v | 0x80); v >>= 7; }
    o[s++] = (uint8_t)v;
    return s;
}
static int get_varint(const uint8_t *b, size_t len, size_t *pos, uint64_t *v) {
    uint64_t r = 0; int shift = 0;
    while (*pos < len && shift < 70) {
        uint8_t c = b[(*pos)++];
        r |= (uint64_t)(c & 0x7f) << shift;
        if (!(c & 0x80)) { *v = r; return 0; }
        shift += 7;
    }
    return 1;
}
static size_t payload_size(const uint64_t *vec, size_t n) { size_t s = 0; for (size_t i = 0; i < n; i++) s += varint_size(vec[i]); return s; }

size_t pmsg_packed_size(const uint64_t *vec, size_t n, uint64_t value) {
    size_t s = 1 + varint_size(value);                       /* field 2, varint */
    if (n) { size_t p = payload_size(vec, n); s += 1 + varint_size(p) + p; }   /* field 1, length-delimited */
    return s;
}
size_t pmsg_pack(const uint64_t *vec, size_t n, uint64_t value, uint8_t *o) {
    size_t s = 0;
    if (n) {
        o[s++] = 0x0a;                                       /* (1 << 3) | 2 */
        s += put_varint(payload_size(vec, n), o + s);
        for (size_t i = 0; i < n; i++) s += put_varint(vec[i], o + s);
    }
    o[s++] = 0x10;                                           /* (2 << 3) | 0 */
    s += put_varint(value, o + s);
    return s;
}
int pmsg_unpack(const uint8_t *b, size_t len, uint64_t **vector, size_t *n, uint64_t *value) {
    size_t pos = 0, cnt = 0, cap = 0;
    uint64_t *vec = 0;
    int have_value = 0;
    while (pos < len) {
        uint64_t key;
        if (get_varint(b, len, &pos, &key)) goto bad;
        if (key == 0x0a) {
            uint64_t plen;
            if (get_varint(b, len, &pos, &plen) || pos + plen > len) goto bad;
            size_t end = pos + (size_t)plen;
            while (pos < end) {
                uint64_t v;
                if (get_varint(b, end, &pos, &v)) goto bad;
                if (cnt == cap) { cap = cap ? cap * 2 : 1024; vec = realloc(vec, cap * sizeof *vec); if (!vec) return 1; }
                vec[cnt++] = v;
            }
        } else if (key == 0x08) {                            /* unpacked repeated element */
            uint64_t v;
            if (get_varint(b, len, &pos, &v)) goto bad;
            if (cnt == cap) { cap = cap ? cap * 2 : 1024; vec = realloc(vec, cap * sizeof *vec); if (!vec) return 1; }
            vec[cnt++] = v;
        } else if (key == 0x10) {
            if (get_varint(b, len, &pos, value)) goto bad;
            have_value = 1;
        } else goto bad;
    }
    if (!have_value) goto bad;                               /* `value` is required */
    *vector = vec; *n = cnt;
    return 0;
bad:
    free(vec);
    return 1;
}
