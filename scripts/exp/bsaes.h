// Experiment (not product code): bitsliced fixed-key AES-128 over 32-bit planes -- 32 blocks per lane, no table, no LDS.
// State: s[32 * c + 8 * r + b] = bit b (0 = least significant) of byte r of column c, one bit per block in each of the 32
// positions of the word (the block layout of gc_aes.h: word c of a block is state column c, byte r of the word is row r).
// S-box: bsaes_sbox.inc (gen_bsaes.py: Boyar-Peralta's depth-16 circuit mapped onto three-input cells = v_bitop3_b32).
// ShiftRows is a renaming; MixColumns is t = a_r ^ a_(r+1) (^ key), out_r = xtime(t_r) ^ a_(r+1) ^ t_(r+2); the round key is
// folded into the t's: kappa with 2 * kappa_r + kappa_(r+2) = rk_r per column (bs_fold_key), so AddRoundKey costs nothing.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BS_HD __host__ __device__ __forceinline__
#else
#define BS_HD inline
#define __device__
#define __host__
#define __forceinline__ inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define BS_LUT(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
#else
static inline uint32_t bs_lut_host(uint32_t a, uint32_t b, uint32_t c, uint32_t tt) {
    uint32_t r = 0;
    for (int i = 0; i < 8; i++)
        if ((tt >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
#define BS_LUT(a, b, c, tt) bs_lut_host((a), (b), (c), (tt))
#endif

#include "bsaes_sbox.inc"

// three-input XOR / bit-field insert (m ? x : y) as cells
static BS_HD uint32_t bs_xor3(uint32_t a, uint32_t b, uint32_t c) { return BS_LUT(a, b, c, 0x96); }
static BS_HD uint32_t bs_bfi(uint32_t m, uint32_t x, uint32_t y) { return BS_LUT(m, x, y, 0xCA); }

// per round 1..9: four words kappa[c] with bit (8 r + b) = the fold of the round key for byte r of column c;
// rounds 0 and 10: the plain round-key words
struct BsKey {
    uint32_t k0[4];        // round 0, plain
    uint32_t kap[9][4];    // rounds 1..9, folded into the MixColumns t's
    uint32_t k10[4];       // last round, plain
};

static inline uint8_t bs_gmul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    for (int i = 0; i < 8; i++) {
        if (b & 1) r ^= a;
        uint8_t h = a & 0x80;
        a = (uint8_t)(a << 1);
        if (h) a ^= 0x1b;
        b >>= 1;
    }
    return r;
}
// out_r = 2 t_r ^ a_(r+1) ^ t_(r+2) with t_r = a_r ^ a_(r+1) ^ kappa_r: the key's contribution is E_r = 2 kappa_r ^ kappa_(r+2).
// Solve E = rk per column: kappa_r = (2 rk_r ^ rk_(r+2)) / 5 (from E_r, E_(r+2): determinant 4 ^ 1 = 5).
static inline void bs_fold_key(const uint32_t rk[44], BsKey &k) {
    uint8_t inv5 = 0;
    for (int x = 1; x < 256; x++)
        if (bs_gmul((uint8_t)x, 5) == 1) inv5 = (uint8_t)x;
    for (int c = 0; c < 4; c++) { k.k0[c] = rk[c]; k.k10[c] = rk[40 + c]; }
    for (int rnd = 1; rnd <= 9; rnd++)
        for (int c = 0; c < 4; c++) {
            uint8_t e[4], kap[4];
            for (int r = 0; r < 4; r++) e[r] = (uint8_t)(rk[4 * rnd + c] >> (8 * r));
            for (int r = 0; r < 4; r++) kap[r] = bs_gmul((uint8_t)(bs_gmul(e[r], 2) ^ e[(r + 2) & 3]), inv5);
            k.kap[rnd - 1][c] = (uint32_t)kap[0] | ((uint32_t)kap[1] << 8) | ((uint32_t)kap[2] << 16) | ((uint32_t)kap[3] << 24);
        }
}

// bit -> all-ones / zero mask of a (wave-uniform) key word: scalar work on the GPU
static BS_HD uint32_t bs_kbit(uint32_t kw, int bit) { return 0u - ((kw >> bit) & 1u); }

static BS_HD void bs_subbytes(uint32_t *s) {
#pragma unroll
    for (int by = 0; by < 16; by++) {
        uint32_t u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = s[8 * by + 7 - i];
        bs_sbox(u);
#pragma unroll
        for (int i = 0; i < 8; i++) s[8 * by + 7 - i] = u[i];
    }
}

// ShiftRows + MixColumns + AddRoundKey (key folded: kap[c])
static BS_HD void bs_mix(uint32_t *s, const uint32_t *kap) {
    uint32_t n[128];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t a[4][8], t[4][8];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int b = 0; b < 8; b++) a[r][b] = s[32 * ((c + r) & 3) + 8 * r + b];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int b = 0; b < 8; b++) t[r][b] = bs_xor3(a[r][b], a[(r + 1) & 3][b], bs_kbit(kap[c], 8 * r + b));
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint32_t hi = t[r][7];
                uint32_t o;
                if (b == 0) o = bs_xor3(hi, a[(r + 1) & 3][0], t[(r + 2) & 3][0]);
                else {
                    o = bs_xor3(t[r][b - 1], a[(r + 1) & 3][b], t[(r + 2) & 3][b]);
                    if (b == 1 || b == 3 || b == 4) o ^= hi;
                }
                n[32 * c + 8 * r + b] = o;
            }
    }
#pragma unroll
    for (int i = 0; i < 128; i++) s[i] = n[i];
}

// last round: ShiftRows + AddRoundKey
static BS_HD void bs_last(uint32_t *s, const uint32_t *k10) {
    uint32_t n[128];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int b = 0; b < 8; b++) n[32 * c + 8 * r + b] = s[32 * ((c + r) & 3) + 8 * r + b] ^ bs_kbit(k10[c], 8 * r + b);
#pragma unroll
    for (int i = 0; i < 128; i++) s[i] = n[i];
}

// 32 x 32 bit-matrix transpose in place: bit j of m[i] <-> bit i of m[j]
static BS_HD void bs_transpose32(uint32_t *m) {
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
        uint32_t mask = (j == 16) ? 0x0000ffffu : (j == 8) ? 0x00ff00ffu : (j == 4) ? 0x0f0f0f0fu : (j == 2) ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t lo = m[k], hi = m[k + j];
            m[k] = bs_bfi(mask, lo, hi << j);
            m[k + j] = bs_bfi(mask, lo >> j, hi);
        }
    }
}

// 32 blocks (w[blk][c]) -> planes, with the round-0 key; and back
static BS_HD void bs_load(uint32_t *s, const uint32_t w[32][4], const uint32_t *k0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t m[32];
#pragma unroll
        for (int j = 0; j < 32; j++) m[j] = w[j][c] ^ k0[c];
        bs_transpose32(m);
#pragma unroll
        for (int j = 0; j < 32; j++) s[32 * c + j] = m[j];
    }
}
static BS_HD void bs_store(const uint32_t *s, uint32_t w[32][4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t m[32];
#pragma unroll
        for (int j = 0; j < 32; j++) m[j] = s[32 * c + j];
        bs_transpose32(m);
#pragma unroll
        for (int j = 0; j < 32; j++) w[j][c] = m[j];
    }
}

// planes (round-0 key already in) -> planes of the ciphertext.  The round loop stays rolled on the GPU: its body is
// ~1 800 instructions (14 KiB of code); unrolled ten times it would not fit the instruction cache.
static BS_HD void bs_encrypt_planes(uint32_t *s, const BsKey &k) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int rnd = 0; rnd < 9; rnd++) {
        bs_subbytes(s);
        bs_mix(s, k.kap[rnd]);
    }
    bs_subbytes(s);
    bs_last(s, k.k10);
}
