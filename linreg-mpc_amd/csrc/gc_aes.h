// gc_aes.h -- fixed-key AES-128 and the garbling hash, host + device.
//
// CDNA4 has no AES instruction: the device path is a T-table AES with the tables staged in LDS
// (layouts in gc_device.h).  A ds_read_b32 is serviced in two groups of 32 lanes over 32 banks
// (MI355X_MICROARCH.md, LDS), so 32 replicas of an entry -- lane l reads replica l mod 32 -- are
// conflict-free whatever the data, and the LDS byte address (entry << 8 | table bits | replica << 2)
// is formed by ONE v_perm_b32 from the state word and a per-lane constant.  The MAC kernels and the
// one-workgroup-per-CU generic kernels keep all four rotated tables Te_t = rotl(Te0, 8t) resident
// (128 KiB): a round is 16 lookups + 8 three-input XORs (v_bitop3_b32), no rotates -- 240 VALU + 160
// ds_read_b32 per block.  Kernels that need two workgroups per CU use Te0 | Te2 in 64 KiB (one
// rotate per column); the single-table image (three v_alignbit per column) remains for the small
// input-label and PRG kernels.  Integer VALU issue is 16 lanes/clk/SIMD on CDNA4, so instruction
// count per block bounds the cipher until the LDS roof (160 lookups x 2 clk per wave) takes over.
//
// Conventions: a 128-bit block / label is 4 little-endian u32 words; word c
// is AES state column c, byte r of the word is state row r -- the byte order
// of FIPS-197 and of x86 AES-NI, so CPU (AES-NI) and GPU agree bit for bit.
//
// Hash (half-gates, Zahur-Rosulek-Evans 2015, fixed-key model):
//     H(x, t) = AES_k(sigma(x) ^ t) ^ sigma(x) ^ t,
//     sigma(xL || xR) = (xL ^ xR) || xL   (a linear orthomorphism)
// This replaces Obliv-C's libgcrypt-based gate hash (reference call site:
// execYaoProtocol, src/cmd/linreg.c:177).
#pragma once
#include <stdint.h>

#ifndef GC_HD
#if defined(__HIPCC__)
#define GC_HD __host__ __device__ __forceinline__
#else
#define GC_HD inline
#endif
#endif

namespace gc {

struct Lbl {
    uint32_t x, y, z, w;
};
GC_HD Lbl lxor(Lbl a, Lbl b) { Lbl r = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; return r; }
GC_HD Lbl lzero() { Lbl r = {0, 0, 0, 0}; return r; }
// m ? a : 0 without a branch
GC_HD Lbl lmask(Lbl a, uint32_t m) { uint32_t k = 0u - (m & 1u); Lbl r = {a.x & k, a.y & k, a.z & k, a.w & k}; return r; }

// ---------------------------------------------------------------- host tables
struct AesTables {
    uint8_t sbox[256];
    uint32_t te0[256];
    uint32_t rk[44];
};

constexpr uint8_t aes_xtime(uint8_t v) { return (uint8_t)((v << 1) ^ ((v >> 7) * 0x1b)); }

// S-box from first principles (multiplicative inverse in GF(2^8) + affine map).  constexpr: the device constants of the
// fixed-key hash are COMPILE-TIME data (gc_device.h c_aes) -- nothing is uploaded when a process starts, and a code
// object is loaded when its first kernel is launched, not because its constants had to be written.
constexpr AesTables aes_make_tables(const uint8_t *key) {
    AesTables t{};
    uint8_t pw[256] = {}, lg[256] = {};
    uint8_t g = 1;
    for (int i = 0; i < 255; i++) { pw[i] = g; lg[g] = (uint8_t)i; g = (uint8_t)(g ^ aes_xtime(g)); }  // generator 3
    for (int x = 0; x < 256; x++) {
        uint8_t inv = x ? pw[(255 - lg[x]) % 255] : 0;
        uint8_t s = inv, r = inv;
        for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        t.sbox[x] = (uint8_t)(s ^ 0x63);
    }
    for (int x = 0; x < 256; x++) {
        uint8_t s = t.sbox[x], s2 = aes_xtime(s), s3 = (uint8_t)(s2 ^ s);
        t.te0[x] = (uint32_t)s2 | ((uint32_t)s << 8) | ((uint32_t)s << 16) | ((uint32_t)s3 << 24);
    }
    for (int i = 0; i < 4; i++)
        t.rk[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) |
                  ((uint32_t)key[4 * i + 3] << 24);
    uint8_t rcon = 1;
    for (int i = 4; i < 44; i++) {
        uint32_t v = t.rk[i - 1];
        if (i % 4 == 0) {
            v = (v >> 8) | (v << 24);  // RotWord (bytes are little-endian in the word)
            v = (uint32_t)t.sbox[v & 0xff] | ((uint32_t)t.sbox[(v >> 8) & 0xff] << 8) |
                ((uint32_t)t.sbox[(v >> 16) & 0xff] << 16) | ((uint32_t)t.sbox[(v >> 24) & 0xff] << 24);
            v ^= rcon;
            rcon = aes_xtime(rcon);
        }
        t.rk[i] = t.rk[i - 4] ^ v;
    }
    return t;
}
inline void aes_build_tables(AesTables &t, const uint8_t key[16]) { t = aes_make_tables(key); }

// the fixed public key of this build (FIPS-197 Appendix B example key)
static constexpr uint8_t kFixedKey[16] = {0x2b, 0x7e, 0x15, 0x16, 0x28, 0xae, 0xd2, 0xa6,
                                          0xab, 0xf7, 0x15, 0x88, 0x09, 0xcf, 0x4f, 0x3c};

// what the device kernels read: round keys, Te0, and rotl24 of the round keys (two-table AES rounds)
struct DevAesConst {
    uint32_t rk[44];
    uint32_t te0[256];
    uint32_t rk24[44];
};
constexpr DevAesConst aes_make_dev_const() {
    AesTables t = aes_make_tables(kFixedKey);
    DevAesConst c{};
    for (int i = 0; i < 44; i++) { c.rk[i] = t.rk[i]; c.rk24[i] = (t.rk[i] << 24) | (t.rk[i] >> 8); }
    for (int i = 0; i < 256; i++) c.te0[i] = t.te0[i];
    return c;
}

GC_HD uint32_t rotl32(uint32_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(v, v, 32 - k);
#else
    return (v << k) | (v >> (32 - k));
#endif
}
// three-input XOR: one v_bitop3_b32 on gfx950 (there is no v_xor3 on gfx9)
GC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
// last round: Te0[x] = (2s, s, s, 3s) holds S[x] in bytes 1 and 2; build the
// output column from four lookups with two byte permutes
GC_HD uint32_t last_lo(uint32_t v1, uint32_t v0) {   // byte0 <- S[i0], byte1 <- S[i1]
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(v1, v0, 0x0c0c0501);
#else
    return ((v0 >> 8) & 0xffu) | (v1 & 0xff00u);
#endif
}
GC_HD uint32_t last_hi(uint32_t v3, uint32_t v2) {   // byte2 <- S[i2], byte3 <- S[i3]
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(v3, v2, 0x05020c0c);
#else
    return (v2 & 0xff0000u) | ((v3 << 16) & 0xff000000u);
#endif
}

// N independent blocks, interleaved round by round (ILP hides LDS latency).
// T: table accessor, T::lk(word, k) returns Te0[byte k of word].
template <int N, class T>
GC_HD void aes_encrypt_n(const T &tab, const uint32_t *rk, uint32_t s[N][4], const uint32_t *rk24 = 0) {
#pragma unroll
    for (int b = 0; b < N; b++) {
        s[b][0] ^= rk[0]; s[b][1] ^= rk[1]; s[b][2] ^= rk[2]; s[b][3] ^= rk[3];
    }
#pragma unroll
    for (int rnd = 1; rnd < 10; rnd++) {
        uint32_t v[N][16];
#pragma unroll
        for (int b = 0; b < N; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (T::kFourTables) {
                    v[b][4 * j + 0] = tab.lkt(0, s[b][j], 0);
                    v[b][4 * j + 1] = tab.lkt(1, s[b][(j + 1) & 3], 1);
                    v[b][4 * j + 2] = tab.lkt(2, s[b][(j + 2) & 3], 2);
                    v[b][4 * j + 3] = tab.lkt(3, s[b][(j + 3) & 3], 3);
                    continue;
                }
                v[b][4 * j + 0] = tab.lk(s[b][j], 0);
                v[b][4 * j + 1] = tab.lk(s[b][(j + 1) & 3], 1);
                v[b][4 * j + 2] = T::kTwoTables ? tab.lk2(s[b][(j + 2) & 3], 2) : tab.lk(s[b][(j + 2) & 3], 2);
                v[b][4 * j + 3] = T::kTwoTables ? tab.lk2(s[b][(j + 3) & 3], 3) : tab.lk(s[b][(j + 3) & 3], 3);
            }
        }
#pragma unroll
        for (int b = 0; b < N; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (T::kFourTables) {
                    // Te_t = rotl(Te0, 8t) are all resident: no rotates
                    s[b][j] = xor3(xor3(v[b][4 * j], v[b][4 * j + 1], rk[4 * rnd + j]), v[b][4 * j + 2], v[b][4 * j + 3]);
                } else if (T::kTwoTables) {
                    // v2, v3 come from Te2 = rotl16(Te0): col = Te0[i0]^Te2[i2] ^ rotl8(Te0[i1]^Te2[i3]) ^ rk,
                    // with the round key folded in before the rotation (rk24 = rotl24(rk))
                    uint32_t x = xor3(v[b][4 * j + 1], v[b][4 * j + 3], rk24[4 * rnd + j]);
                    s[b][j] = xor3(v[b][4 * j], v[b][4 * j + 2], rotl32(x, 8));
                } else {
                    uint32_t t = xor3(v[b][4 * j], rotl32(v[b][4 * j + 2], 16), rk[4 * rnd + j]);
                    s[b][j] = xor3(t, rotl32(v[b][4 * j + 1], 8), rotl32(v[b][4 * j + 3], 24));
                }
            }
        }
    }
    {
        uint32_t v[N][16];
#pragma unroll
        for (int b = 0; b < N; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[b][4 * j + 0] = tab.lk(s[b][j], 0);
                v[b][4 * j + 1] = tab.lk(s[b][(j + 1) & 3], 1);
                v[b][4 * j + 2] = tab.lk(s[b][(j + 2) & 3], 2);
                v[b][4 * j + 3] = tab.lk(s[b][(j + 3) & 3], 3);
            }
        }
#pragma unroll
        for (int b = 0; b < N; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                s[b][j] = xor3(last_lo(v[b][4 * j + 1], v[b][4 * j]), last_hi(v[b][4 * j + 3], v[b][4 * j + 2]), rk[40 + j]);
        }
    }
}

// K = sigma(x) ^ tweak, as 4 words
GC_HD void hash_prep(Lbl x, uint64_t tweak, uint32_t k[4]) {
    k[0] = x.z ^ (uint32_t)tweak;
    k[1] = x.w ^ (uint32_t)(tweak >> 32);
    k[2] = x.z ^ x.x;
    k[3] = x.w ^ x.y;
}

// N hashes at once: out[b] = AES(K_b) ^ K_b
template <int N, class T>
GC_HD void hash_n(const T &tab, const uint32_t *rk, const Lbl *x, const uint64_t *tw, Lbl *out, const uint32_t *rk24 = 0) {
    uint32_t s[N][4], k[N][4];
#pragma unroll
    for (int b = 0; b < N; b++) {
        hash_prep(x[b], tw[b], k[b]);
        s[b][0] = k[b][0]; s[b][1] = k[b][1]; s[b][2] = k[b][2]; s[b][3] = k[b][3];
    }
    aes_encrypt_n<N, T>(tab, rk, s, rk24);
#pragma unroll
    for (int b = 0; b < N; b++) {
        out[b].x = s[b][0] ^ k[b][0]; out[b].y = s[b][1] ^ k[b][1];
        out[b].z = s[b][2] ^ k[b][2]; out[b].w = s[b][3] ^ k[b][3];
    }
}

// host table accessor (plain array)
struct HostTab {
    static const bool kTwoTables = false;
    static const bool kFourTables = false;
    const uint32_t *te0;
    inline uint32_t lkt(int, uint32_t word, int k) const { return lk(word, k); }
    inline uint32_t lk(uint32_t word, int k) const { return te0[(word >> (8 * k)) & 0xffu]; }
    inline uint32_t lk2(uint32_t word, int k) const { return rotl32(lk(word, k), 16); }
};

// ---- half-gates, one AND gate (lane-local).  gid: unique gate id.
// Garbler: a0, b0 zero-labels; returns c0 and the two ciphertexts.
template <class T>
GC_HD Lbl garble_and(const T &tab, const uint32_t *rk, Lbl R, Lbl a0, Lbl b0, uint64_t gid, Lbl &TG, Lbl &TE,
                     const uint32_t *rk24 = 0) {
    Lbl in[4] = {a0, lxor(a0, R), b0, lxor(b0, R)};
    uint64_t tw[4] = {2 * gid, 2 * gid, 2 * gid + 1, 2 * gid + 1};
    Lbl h[4];
    // two pairs of interleaved blocks (four at once need more live registers than four waves per SIMD leave)
    hash_n<2, T>(tab, rk, in, tw, h, rk24);
    hash_n<2, T>(tab, rk, in + 2, tw + 2, h + 2, rk24);
    uint32_t pa = a0.x & 1u, pb = b0.x & 1u;
    TG = lxor(lxor(h[0], h[1]), lmask(R, pb));
    Lbl WG = lxor(h[0], lmask(TG, pa));
    TE = lxor(lxor(h[2], h[3]), a0);
    Lbl WE = lxor(h[2], lmask(lxor(TE, a0), pb));
    return lxor(WG, WE);
}
// Evaluator: a, b active labels.
template <class T>
GC_HD Lbl eval_and(const T &tab, const uint32_t *rk, Lbl a, Lbl b, uint64_t gid, Lbl TG, Lbl TE, const uint32_t *rk24 = 0) {
    Lbl in[2] = {a, b};
    uint64_t tw[2] = {2 * gid, 2 * gid + 1};
    Lbl h[2];
    hash_n<2, T>(tab, rk, in, tw, h, rk24);
    uint32_t sa = a.x & 1u, sb = b.x & 1u;
    Lbl WG = lxor(h[0], lmask(TG, sa));
    Lbl WE = lxor(h[1], lmask(lxor(TE, a), sb));
    return lxor(WG, WE);
}

}  // namespace gc
