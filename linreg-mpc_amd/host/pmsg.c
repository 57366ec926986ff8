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

/* BMI2 fast paths (x86-64; chosen at run time): a varint is the value's 7-bit groups spread over
 * bytes, which is one pdep (encode) / pext (decode) for the low 56 bits. */
#if defined(__x86_64__)
#include <immintrin.h>
#include <string.h>
#define PMSG_FAST 1
static int have_bmi2(void) {
    static int v = -1;
    if (v < 0) v = __builtin_cpu_supports("bmi2") ? 1 : 0;
    return v;
}
/* writes up to 16 bytes at o (the caller leaves slack), returns the encoded length */
__attribute__((target("bmi2"))) static inline size_t put_varint_bmi2(uint64_t v, uint8_t *o) {
    const unsigned bits = 64u - (unsigned)__builtin_clzll(v | 1);
    const unsigned s = (bits + 6u) / 7u;
    uint64_t lo = _pdep_u64(v, 0x7f7f7f7f7f7f7f7full);
    if (s <= 8) {
        lo |= 0x8080808080808080ull & (((uint64_t)1 << (8u * (s - 1u))) - 1u);
        memcpy(o, &lo, 8);
    } else {
        lo |= 0x8080808080808080ull;
        uint16_t hi = (uint16_t)_pdep_u64(v >> 56, 0x017full);
        if (s == 10) hi |= 0x80;
        memcpy(o, &lo, 8);
        memcpy(o + 8, &hi, 2);
    }
    return s;
}
/* encodes vec[0..n) while a full 10-byte store still fits below `room`; *done = values consumed */
__attribute__((target("bmi2"))) static size_t pack_payload_bmi2(const uint64_t *vec, size_t n, uint8_t *o, size_t room, size_t *done) {
    size_t s = 0, i = 0;
    for (; i < n && s + 10 <= room; i++) s += put_varint_bmi2(vec[i], o + s);
    *done = i;
    return s;
}
/* decodes varints from b[*pos..end) into vec while 10 readable bytes remain below `len`; returns
 * the count or (size_t)-1 on a malformed varint; *pos is advanced */
__attribute__((target("bmi2"))) static size_t unpack_payload_bmi2(const uint8_t *b, size_t *ppos, size_t end, size_t len, uint64_t *vec, size_t maxcnt) {
    size_t cnt = 0, pos = *ppos;
    while (pos < end && pos + 10 <= len) {
        if (cnt == maxcnt) return (size_t)-1;
        uint64_t x;
        memcpy(&x, b + pos, 8);
        uint64_t stop = ~x & 0x8080808080808080ull;
        if (stop) {
            unsigned len = ((unsigned)__builtin_ctzll(stop) >> 3) + 1u;
            uint64_t keep = len == 8 ? ~0ull : (((uint64_t)1 << (8u * len)) - 1u);
            vec[cnt++] = _pext_u64(x & keep, 0x7f7f7f7f7f7f7f7full);
            pos += len;
        } else {
            uint64_t v = _pext_u64(x, 0x7f7f7f7f7f7f7f7full);
            uint8_t c8 = b[pos + 8];
            v |= (uint64_t)(c8 & 0x7f) << 56;
            if (c8 & 0x80) {
                uint8_t c9 = b[pos + 9];
                if (c9 & 0x80) return (size_t)-1;
                v |= (uint64_t)c9 << 63;
                pos += 10;
            } else pos += 9;
            vec[cnt++] = v;
        }
        if (pos > end) return (size_t)-1;
    }
    *ppos = pos;
    return cnt;
}
#else
#define PMSG_FAST 0
#endif

static size_t payload_size(const uint64_t *vec, size_t n) {
    size_t s = 0;
    for (size_t i = 0; i < n; i++) s += (64u - (unsigned)__builtin_clzll(vec[i] | 1) + 6u) / 7u;
    return s;
}

size_t pmsg_packed_size(const uint64_t *vec, size_t n, uint64_t value) {
    size_t s = 1 + varint_size(value);                       /* field 2, varint */
    if (n) { size_t p = payload_size(vec, n); s += 1 + varint_size(p) + p; }   /* field 1, length-delimited */
    return s;
}
size_t pmsg_pack(const uint64_t *vec, size_t n, uint64_t value, uint8_t *o) {
    size_t s = 0;
    if (n) {
        o[s++] = 0x0a;                                       /* (1 << 3) | 2 */
        const size_t plen = payload_size(vec, n);
        s += put_varint(plen, o + s);
        size_t i = 0;
#if PMSG_FAST
        if (have_bmi2()) s += pack_payload_bmi2(vec, n, o + s, plen, &i);
#endif
        for (; i < n; i++) s += put_varint(vec[i], o + s);
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
#if PMSG_FAST
            if (have_bmi2() && plen >= 64) {                 /* every varint is at least one byte */
                if (cnt + (size_t)plen > cap) { cap = cnt + (size_t)plen; vec = realloc(vec, cap * sizeof *vec); if (!vec) return 1; }
                size_t got = unpack_payload_bmi2(b, &pos, end, len, vec + cnt, (size_t)plen);
                if (got == (size_t)-1) goto bad;
                cnt += got;
            }
#endif
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

/* as pmsg_unpack, into a caller-owned vector of `cap` words (no allocation: the phase-1 workers decode
 * straight into page-locked batch buffers); more than cap elements is an error */
int pmsg_unpack_into(const uint8_t *b, size_t len, uint64_t *vec, size_t cap, size_t *n, uint64_t *value) {
    size_t pos = 0, cnt = 0;
    int have_value = 0;
    while (pos < len) {
        uint64_t key;
        if (get_varint(b, len, &pos, &key)) return 1;
        if (key == 0x0a) {
            uint64_t plen;
            if (get_varint(b, len, &pos, &plen) || pos + plen > len) return 1;
            size_t end = pos + (size_t)plen;
#if PMSG_FAST
            if (have_bmi2() && plen >= 64) {
                size_t got = unpack_payload_bmi2(b, &pos, end, len, vec + cnt, cap - cnt);
                if (got == (size_t)-1) return 1;
                cnt += got;
            }
#endif
            while (pos < end) {
                uint64_t v;
                if (cnt == cap || get_varint(b, end, &pos, &v)) return 1;
                vec[cnt++] = v;
            }
        } else if (key == 0x08) {
            uint64_t v;
            if (cnt == cap || get_varint(b, len, &pos, &v)) return 1;
            vec[cnt++] = v;
        } else if (key == 0x10) {
            if (get_varint(b, len, &pos, value)) return 1;
            have_value = 1;
        } else return 1;
    }
    if (!have_value) return 1;
    *n = cnt;
    return 0;
}
