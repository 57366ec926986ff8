// ot.hip -- IKNP OT extension (semi-honest) on the MI355X.
//
// Replaces Obliv-C's honestOTExt* / honestCorrelatedOTExt* as called by the reference at
//   src/phase1.c:58-65, 84-89   Gilboa inner products over correlated OT (one OT per bit of a_k)
//   src/input.c:44, 108         1-of-2 OT of wire labels (data provider <- CSP)
// The kappa = 128 base OTs (Naor-Pinkas in Obliv-C, dhRandomInit) are public-key work on the
// host and not part of this path: the entry points take their outputs (seeds).
//
// Protocol (Ishai-Kilian-Nissim-Petrank 2003, with the correlated-OT optimisation of
// Asharov-Lindell-Schneider-Zohner 2013), m OTs, choice vector c:
//   receiver: t_j = G(k_j^0), u_j = t_j ^ G(k_j^1) ^ c            j = 0..127   (sends u: 16 B / OT)
//   sender  : q_j = G(k_j^{D_j}) ^ D_j * u_j       => row i: q_i = t_i ^ c_i * D
//   G       : AES-128-CTR keyed by the seed (per-column key schedule)
//   rows    : 128 x m bit-matrix transpose (wave ballots)
//   hash    : H(i, x) = the fixed-key hash of gc_aes.h with tweak i (unique per session)
//   correlated (Gilboa, payload w bits): x0_i = H(i, q_i), y_i = x0_i + delta_i - H(i, q_i ^ D);
//             receiver gets H(i, t_i) + c_i * y_i.  delta_i = 2^bit * b_k  (phase1.c:43-51)
//   labels  : e0_i = m0_i ^ H(i, q_i), e1_i = m1_i ^ H(i, q_i ^ D); receiver m_c = e_c ^ H(i, t_i)
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"
#include "hip_scope.h"
#include "gc_device.h"

using namespace gc;

int lgc_fail(int code, const char *fmt, ...);
int lgc_need_device(int device);
hipError_t lgc_stream_take(int device, hipStream_t *out);      // gc_engine.hip: pooled streams (hipStreamCreate is ~10 ms)
void lgc_stream_give(int device, hipStream_t st);
__global__ void ot_tu_touch_kernel() {}
hipError_t ot_tu_touch(hipStream_t st) { hipLaunchKernelGGL(ot_tu_touch_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }

#define OTCHK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) return lgc_fail(LGC_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- column PRG.  grid.y = column j (0..127); each lane produces AES_kj(ctr0 + b) for block b.
// MODE 0 (receiver): T0[j][b] = G(k0), U[j][b] = G(k0) ^ G(k1) ^ c[b]
// MODE 1 (sender)  : Q[j][b]  = G(k)  ^ (delta_j ? U[j][b] : 0)
// Four-table AES image (128 KiB, one workgroup per CU) with the column's own key schedule in LDS.
template <int MODE>
__global__ void __launch_bounds__(1024)
ot_cols_kernel(const uint32_t *rk0, const uint32_t *rk1, uint64_t ctr0, uint32_t m128, const uint4 *cbits,
               const uint4 *Uin, uint4 delta, uint4 *out0, uint4 *out1) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    __shared__ uint32_t sk0[44], sk1[44];
    lds_tab4_fill(lds_te0);
    const uint32_t j = blockIdx.y;
    if (threadIdx.x < 44) {
        sk0[threadIdx.x] = rk0[j * 44 + threadIdx.x];
        if (MODE == 0) sk1[threadIdx.x] = rk1[j * 44 + threadIdx.x];
    }
    __syncthreads();
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint32_t dw = j < 32 ? delta.x : (j < 64 ? delta.y : (j < 96 ? delta.z : delta.w));
    const bool dj = (dw >> (j & 31)) & 1u;
    // two counter blocks per trip (two AES in flight per key); the grid is a few workgroups per COLUMN, not per 1024
    // blocks: every workgroup stages the 128 KiB table image once (~25 us), which with 64 workgroups per column cost
    // more than the encryption itself
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < m128; b += 2 * stride) {
        const uint32_t b2 = b + stride;
        const bool two = b2 < m128;
        const uint64_t c = ctr0 + b, c2 = ctr0 + (two ? b2 : b);
        uint32_t s0[2][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}, {(uint32_t)c2, (uint32_t)(c2 >> 32), 0u, 0u}};
        aes_encrypt_n<2, LdsTab4>(lt, sk0, s0);
        if (MODE == 0) {
            uint32_t s1[2][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}, {(uint32_t)c2, (uint32_t)(c2 >> 32), 0u, 0u}};
            aes_encrypt_n<2, LdsTab4>(lt, sk1, s1);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (e == 1 && !two) break;
                const uint32_t bb = e ? b2 : b;
                uint4 cb = cbits[bb];
                out0[(size_t)j * m128 + bb] = make_uint4(s0[e][0], s0[e][1], s0[e][2], s0[e][3]);
                out1[(size_t)j * m128 + bb] = make_uint4(s0[e][0] ^ s1[e][0] ^ cb.x, s0[e][1] ^ s1[e][1] ^ cb.y,
                                                         s0[e][2] ^ s1[e][2] ^ cb.z, s0[e][3] ^ s1[e][3] ^ cb.w);
            }
        } else {
            const uint32_t k = dj ? 0xffffffffu : 0u;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (e == 1 && !two) break;
                const uint32_t bb = e ? b2 : b;
                uint4 u = Uin[(size_t)j * m128 + bb];
                out0[(size_t)j * m128 + bb] = make_uint4(s0[e][0] ^ (u.x & k), s0[e][1] ^ (u.y & k), s0[e][2] ^ (u.z & k),
                                                         s0[e][3] ^ (u.w & k));
            }
        }
    }
}

// ---- 128 x m bit transpose: cols[j][m128] (bit i of column j = bit (i & 127) of block i >> 7)
// -> rows[i] = 128 bits (bit j = column j).  One wave per 64 consecutive OTs: lane l loads the 64-bit
// word of column l (and l + 64) that covers them, and the wave transposes the 64 x 64 bit block in
// registers -- six butterfly stages, lane l exchanging with lane l ^ j (ds_bpermute) and swapping the
// off-diagonal j x j sub-blocks -- instead of one ballot per bit.
__device__ __forceinline__ uint64_t transpose64_lanes(uint64_t x, int lane) {
    uint64_t m = 0x00000000ffffffffull;
#pragma unroll
    for (int j = 32; j != 0; j >>= 1, m ^= (m << j)) {
        const int src = (lane ^ j) << 2;
        const uint32_t ylo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)x);
        const uint32_t yhi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(x >> 32));
        const uint64_t y = ((uint64_t)yhi << 32) | ylo;
        // row k (bit j of the lane clear) keeps its low part and takes the partner's low part as its high part;
        // row k + j keeps its high part and takes the partner's high part as its low part
        x = (lane & j) ? ((x & ~m) | ((y >> j) & m)) : ((x & m) | ((y << j) & ~m));
    }
    return x;
}
// One wave transposes EIGHT consecutive 64-OT groups: lane l then reads 64 contiguous bytes of column l (and of column
// 64 + l) -- whole 64-byte sectors.  With one group per wave a lane read 8 bytes per column, each from a cache line of
// its own (the columns are m / 8 bytes apart): an eighth of every sector fetched was used, and the transpose, not the
// hashing, was the longest kernel of a batch (profiles/r3b_ot_kernel_stats.csv: 2 x 0.70 ms of 4.55 ms).
__global__ void __launch_bounds__(256)
ot_transpose_kernel(const uint64_t *cols, uint32_t m128, uint4 *rows, uint64_t m) {
    const int lane = threadIdx.x & 63;
    const uint64_t wv = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t i0 = wv * 512;
    if (i0 >= m) return;
    const uint64_t w64 = (uint64_t)m128 * 2;                       // 64-bit words per column
    const uint64_t first = i0 >> 6;                                // multiple of 8: 64-byte aligned within a column
    uint64_t a[8], b[8];
    const uint64_t *c0 = cols + (uint64_t)lane * w64 + first, *c1 = cols + (uint64_t)(64 + lane) * w64 + first;
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
        if (first + t + 1 < w64) {                                 // (w64 is even: pairs never straddle the end)
            const uint4 u = *reinterpret_cast<const uint4 *>(c0 + t), v = *reinterpret_cast<const uint4 *>(c1 + t);
            a[t] = (uint64_t)u.x | ((uint64_t)u.y << 32); a[t + 1] = (uint64_t)u.z | ((uint64_t)u.w << 32);
            b[t] = (uint64_t)v.x | ((uint64_t)v.y << 32); b[t + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        } else {
            a[t] = a[t + 1] = b[t] = b[t + 1] = 0;
        }
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint64_t i = i0 + 64 * (uint64_t)t + (uint64_t)lane;
        if (i0 + 64 * (uint64_t)t >= m) break;                     // wave-uniform
        const uint64_t r0 = transpose64_lanes(a[t], lane), r1 = transpose64_lanes(b[t], lane);
        if (i < m) rows[i] = make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32));
    }
}

__device__ __forceinline__ Lbl u4_lbl(uint4 v) { Lbl l = {v.x, v.y, v.z, v.w}; return l; }

// ---- Gilboa sender: OT i = (q * n + k) * w + bit.  y_i and per-pair share -sum x0.
// grid.y strides over the pairs (gridDim.y is limited to 65535).
__global__ void __launch_bounds__(1024)
ot_gilboa_send_kernel(const uint4 *rows, uint4 delta, const uint64_t *bvals, uint64_t n, int w, uint64_t m_per_pair,
                      uint64_t npairs, uint64_t tweak0, uint64_t *y, uint64_t *shares) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint64_t mask = w == 32 ? 0xffffffffull : ~0ull;
    const int lw = w == 32 ? 5 : 6;
    for (uint64_t q = blockIdx.y; q < npairs; q += gridDim.y) {
        uint64_t acc = 0;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        // two OTs per trip: four hashes, two in flight at a time (the pattern of the MAC kernel's garble_and)
        for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m_per_pair; t += 2 * stride) {
            const uint64_t t2 = t + stride;
            const bool two = t2 < m_per_pair;
            const uint64_t i = q * m_per_pair + t, i2 = q * m_per_pair + (two ? t2 : t);
            const Lbl r1 = u4_lbl(rows[i]), r2 = u4_lbl(rows[i2]);
            Lbl x[4] = {r1, lxor(r1, u4_lbl(delta)), r2, lxor(r2, u4_lbl(delta))};
            uint64_t tw[4] = {tweak0 + i, tweak0 + i, tweak0 + i2, tweak0 + i2};
            Lbl h[4];
            hash_n<2, LdsTab4>(lt, c_aes.rk, x, tw, h, c_aes.rk24);
            hash_n<2, LdsTab4>(lt, c_aes.rk, x + 2, tw + 2, h + 2, c_aes.rk24);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (e == 1 && !two) break;
                const uint64_t tt = e ? t2 : t, ii = e ? i2 : i;
                const uint64_t k = tt >> lw;            // w is 32 or 64: no 64-bit division in the inner loop
                const int bit = (int)(tt & (uint64_t)(w - 1));
                uint64_t x0 = ((uint64_t)h[2 * e].x | ((uint64_t)h[2 * e].y << 32)) & mask;
                uint64_t h1 = ((uint64_t)h[2 * e + 1].x | ((uint64_t)h[2 * e + 1].y << 32)) & mask;
                uint64_t d = (bvals[q * n + k] << bit) & mask;
                y[ii] = (x0 + d - h1) & mask;
                acc -= x0;
            }
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&shares[q], (unsigned long long)acc);
    }
}

__global__ void __launch_bounds__(1024)
ot_gilboa_recv_kernel(const uint4 *rows, const uint64_t *avals, uint64_t n, int w, uint64_t m_per_pair, uint64_t npairs,
                      uint64_t tweak0, const uint64_t *y, uint64_t *shares) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint64_t mask = w == 32 ? 0xffffffffull : ~0ull;
    const int lw = w == 32 ? 5 : 6;
    for (uint64_t q = blockIdx.y; q < npairs; q += gridDim.y) {
        uint64_t acc = 0;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m_per_pair; t += 2 * stride) {
            const uint64_t t2 = t + stride;
            const bool two = t2 < m_per_pair;
            const uint64_t i = q * m_per_pair + t, i2 = q * m_per_pair + (two ? t2 : t);
            Lbl x[2] = {u4_lbl(rows[i]), u4_lbl(rows[i2])};
            uint64_t tw[2] = {tweak0 + i, tweak0 + i2};
            const uint64_t yv[2] = {y[i], y[i2]};         // in flight during the hashes
            Lbl h[2];
            hash_n<2, LdsTab4>(lt, c_aes.rk, x, tw, h, c_aes.rk24);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (e == 1 && !two) break;
                const uint64_t tt = e ? t2 : t;
                const uint64_t k = tt >> lw;            // w is 32 or 64: no 64-bit division in the inner loop
                const int bit = (int)(tt & (uint64_t)(w - 1));
                uint64_t v = ((uint64_t)h[e].x | ((uint64_t)h[e].y << 32)) & mask;
                if ((avals[q * n + k] >> bit) & 1ull) v += yv[e];
                acc += v;
            }
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&shares[q], (unsigned long long)acc);
    }
}

// choice bits of the Gilboa receiver: bit i = bit (i % w) of a[i / w], packed LSB-first
__global__ void ot_pack_choice_words_kernel(const uint64_t *avals, uint64_t nwords, int w, uint64_t *cbits) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 64) {
        if (t < nwords) cbits[t] = avals[t];
    } else {
        if (2 * t < nwords) {
            uint64_t lo = avals[2 * t] & 0xffffffffull;
            uint64_t hi = 2 * t + 1 < nwords ? (avals[2 * t + 1] & 0xffffffffull) : 0;
            cbits[t] = lo | (hi << 32);
        }
    }
}
// choice bits of a label transfer: one byte per OT (bool *sel, src/input.c:40-44) -> packed LSB-first
__global__ void ot_pack_choice_bytes_kernel(const uint8_t *choice, uint64_t m, uint64_t *cbits, uint64_t nwords64) {
    const uint64_t wv = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wv >= nwords64) return;
    const uint64_t i = wv * 64 + lane;
    const uint64_t bits = __ballot(i < m && choice[i] != 0);
    if (lane == 0) cbits[wv] = bits;
}

// ---- 1-of-2 OT of 16-byte messages
__global__ void __launch_bounds__(1024)
ot_labels_send_kernel(const uint4 *rows, uint4 delta, const uint4 *m0, const uint4 *m1, uint64_t m, uint64_t tweak0, uint4 *e) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += 2 * stride) {
        const bool two = i + stride < m;
        const uint64_t i2 = two ? i + stride : i;
        const Lbl r1 = u4_lbl(rows[i]), r2 = u4_lbl(rows[i2]);
        Lbl x[4] = {r1, lxor(r1, u4_lbl(delta)), r2, lxor(r2, u4_lbl(delta))};
        uint64_t tw[4] = {tweak0 + i, tweak0 + i, tweak0 + i2, tweak0 + i2};
        Lbl h[4];
        hash_n<2, LdsTab4>(lt, c_aes.rk, x, tw, h, c_aes.rk24);
        hash_n<2, LdsTab4>(lt, c_aes.rk, x + 2, tw + 2, h + 2, c_aes.rk24);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !two) break;
            const uint64_t ii = k ? i2 : i;
            uint4 a = m0[ii], b = m1[ii];
            e[2 * ii] = make_uint4(a.x ^ h[2 * k].x, a.y ^ h[2 * k].y, a.z ^ h[2 * k].z, a.w ^ h[2 * k].w);
            e[2 * ii + 1] = make_uint4(b.x ^ h[2 * k + 1].x, b.y ^ h[2 * k + 1].y, b.z ^ h[2 * k + 1].z, b.w ^ h[2 * k + 1].w);
        }
    }
}
__global__ void __launch_bounds__(1024)
ot_labels_recv_kernel(const uint4 *rows, const uint64_t *cbits, const uint4 *e, uint64_t m, uint64_t tweak0, uint4 *out) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += 2 * stride) {
        const bool two = i + stride < m;
        const uint64_t i2 = two ? i + stride : i;
        Lbl x[2] = {u4_lbl(rows[i]), u4_lbl(rows[i2])};
        uint64_t tw[2] = {tweak0 + i, tweak0 + i2};
        const uint32_t c1 = (uint32_t)(cbits[i >> 6] >> (i & 63)) & 1u, c2 = (uint32_t)(cbits[i2 >> 6] >> (i2 & 63)) & 1u;
        const uint4 ev1 = e[2 * i + c1], ev2 = e[2 * i2 + c2];      // in flight during the hashes
        Lbl h[2];
        hash_n<2, LdsTab4>(lt, c_aes.rk, x, tw, h, c_aes.rk24);
        out[i] = make_uint4(ev1.x ^ h[0].x, ev1.y ^ h[0].y, ev1.z ^ h[0].z, ev1.w ^ h[0].w);
        if (two) out[i2] = make_uint4(ev2.x ^ h[1].x, ev2.y ^ h[1].y, ev2.z ^ h[1].z, ev2.w ^ h[1].w);
    }
}

// Launch geometry.  Every workgroup of these kernels stages the 128 KiB four-table AES image in LDS before its first
// block (one workgroup per CU, ~25 us), so a launch uses about two workgroups per CU and lets them loop -- not one
// workgroup per 1024 items: with 16 384 workgroups for a 64-pair batch the staging alone took 1.7 ms of each 2.1 ms
// payload kernel (profiles/r2e_ot_kernel_stats.csv; after: profiles/r3_ot_kernel_stats.csv).
static const unsigned kOtGroups = 512;
static unsigned ot_cols_groups(uint64_t m128) {           // workgroups per column (grid.y = 128 columns)
    unsigned gx = (unsigned)((m128 + 2047) / 2048);
    const unsigned cap = kOtGroups / 128;
    return gx > cap ? cap : (gx ? gx : 1);
}
static void ot_pair_grid(uint64_t mpp, uint64_t npairs, unsigned &gx, unsigned &gy) {   // grid.y strides over the pairs
    gy = npairs > kOtGroups ? kOtGroups : (unsigned)(npairs ? npairs : 1);
    gx = (unsigned)((mpp + 2047) / 2048);
    const unsigned cap = kOtGroups / gy ? kOtGroups / gy : 1;
    if (gx > cap) gx = cap;
    if (!gx) gx = 1;
}

// =============================================================== sessions
// Every session owns a HIP stream and grow-only device buffers: no hipMalloc / hipFree on the transfer
// paths after the first call of a given size.  Host arguments move with hipMemcpyAsync on the session
// stream (full PCIe rate for page-locked buffers, lgc_host_alloc); with device I/O switched on
// (lgc_ot_*_set_device_io) the pointer arguments ARE device memory and the kernels use them in place --
// the hand-off for callers that already hold the operands in HBM (lgc_p1_* outputs, a peer's u / y
// mapped over hipIpc), without a host round trip.
struct DevBuf {
    void *p;
    size_t cap;
    DevBuf() : p(0), cap(0) {}
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipMemset(p, 0, cap); (void)hipFree(p); }   // q-rows / messages never stay behind in freed memory
        p = 0; cap = 0;
        size_t want = bytes + bytes / 8;                  // headroom: batches of slightly different size do not reallocate
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = 0; cap = 0; }
};
struct lgc_ot_sender {
    int device;
    uint4 delta;
    uint32_t *rk;          // 128 x 44 round keys of G(k_j^{Delta_j})
    uint64_t ctr;          // PRG stream position (blocks), same on both sides
    uint64_t tweak;        // hash tweak counter (OT index), same on both sides
    hipStream_t st;
    bool dev_io;
    DevBuf Q, U, rows, b, y, sh, m0, m1, e;
};
static const size_t kMaxRecvInFlight = 4;
struct lgc_ot_receiver {
    int device;
    uint32_t *rk0, *rk1;
    uint64_t ctr, tweak;
    hipStream_t st;
    bool dev_io;
    DevBuf T0, U, y, sh, e, out, choice;
    // receives started but not finished yet, oldest first: *_recv_start fills the next slot and
    // *_recv_finish takes the oldest, so a caller may keep a few batches in flight (one thread
    // starting and sending u, another receiving the replies and finishing)
    struct Slot {
        DevBuf rows, cbits, avals;
        const uint64_t *avals_dev;     // device I/O: the caller's a stays where it is
        uint64_t m, npairs, n; int w; uint64_t tweak_cur; bool gilboa;
    } slot[kMaxRecvInFlight];
    size_t head, count;                // ring of slots in use
    std::mutex mu;
};

static int upload_keys(const uint8_t seeds[][16], uint32_t **out) {
    std::vector<uint32_t> rk(128 * 44);
    for (int j = 0; j < 128; j++) {
        AesTables t;
        aes_build_tables(t, seeds[j]);
        memcpy(&rk[j * 44], t.rk, sizeof(t.rk));
    }
    OTCHK(hipMalloc(out, rk.size() * 4));
    OTCHK(hipMemcpy(*out, rk.data(), rk.size() * 4, hipMemcpyHostToDevice));
    return LGC_OK;
}

extern "C" void lgc_ot_sender_destroy(lgc_ot_sender *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->st) { (void)hipStreamSynchronize(s->st); lgc_stream_give(s->device, s->st); }
    // q-rows and the sender's messages (both labels of every input bit) must not outlive the session in freed memory
    DevBuf *all[] = {&s->Q, &s->U, &s->rows, &s->b, &s->y, &s->sh, &s->m0, &s->m1, &s->e};
    for (DevBuf *d : all) { if (d->p) (void)hipMemset(d->p, 0, d->cap); d->release(); }
    if (s->rk) (void)hipFree(s->rk);
    delete s;
}
extern "C" int lgc_ot_sender_create(lgc_ot_sender **out, int device, const uint8_t delta[16], const uint8_t seeds[128][16]) {
    if (!out || !delta || !seeds) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    lgc_ot_sender *s = new lgc_ot_sender();
    s->device = device; s->ctr = 0; s->tweak = 0; s->rk = 0; s->st = 0; s->dev_io = false;
    memcpy(&s->delta, delta, 16);
    rc = upload_keys(seeds, &s->rk);
    if (rc) { lgc_ot_sender_destroy(s); return rc; }
    *out = s;
    return LGC_OK;
}
extern "C" void lgc_ot_receiver_destroy(lgc_ot_receiver *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->st) { (void)hipStreamSynchronize(r->st); lgc_stream_give(r->device, r->st); }
    DevBuf *all[] = {&r->T0, &r->U, &r->y, &r->sh, &r->e, &r->out, &r->choice};
    for (DevBuf *d : all) { if (d->p) (void)hipMemset(d->p, 0, d->cap); d->release(); }
    for (size_t k = 0; k < kMaxRecvInFlight; k++) {
        DevBuf *sl[] = {&r->slot[k].rows, &r->slot[k].cbits, &r->slot[k].avals};
        for (DevBuf *d : sl) { if (d->p) (void)hipMemset(d->p, 0, d->cap); d->release(); }
    }
    if (r->rk0) (void)hipFree(r->rk0);
    if (r->rk1) (void)hipFree(r->rk1);
    delete r;
}
extern "C" int lgc_ot_receiver_create(lgc_ot_receiver **out, int device, const uint8_t seeds0[128][16], const uint8_t seeds1[128][16]) {
    if (!out || !seeds0 || !seeds1) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    lgc_ot_receiver *r = new lgc_ot_receiver();
    r->device = device; r->rk0 = 0; r->rk1 = 0; r->ctr = 0; r->tweak = 0; r->st = 0; r->dev_io = false; r->head = 0; r->count = 0;
    rc = upload_keys(seeds0, &r->rk0);
    if (!rc) rc = upload_keys(seeds1, &r->rk1);
    if (rc) { lgc_ot_receiver_destroy(r); return rc; }
    *out = r;
    return LGC_OK;
}
extern "C" int lgc_ot_sender_set_device_io(lgc_ot_sender *s, int on) {
    if (!s) return lgc_fail(LGC_EINVAL, "null session");
    s->dev_io = on != 0;
    return LGC_OK;
}
extern "C" int lgc_ot_receiver_set_device_io(lgc_ot_receiver *r, int on) {
    if (!r) return lgc_fail(LGC_EINVAL, "null session");
    r->dev_io = on != 0;
    return LGC_OK;
}

static inline uint64_t round128(uint64_t m) { return (m + 127) / 128 * 128; }
#define OTLAUNCH() OTCHK(hipGetLastError())

// receiver: columns + u + transpose for the choice bits in slot->cbits (padded to m128 blocks)
static int recv_extend(lgc_ot_receiver *r, lgc_ot_receiver::Slot *sl, uint64_t m, uint8_t *u_out) {
    const uint32_t m128 = (uint32_t)(round128(m) / 128);
    const size_t cbytes = (size_t)128 * m128 * 16;
    OTCHK(r->T0.ensure(cbytes));
    uint4 *U = reinterpret_cast<uint4 *>(u_out);
    if (!r->dev_io) { OTCHK(r->U.ensure(cbytes)); U = static_cast<uint4 *>(r->U.p); }
    OTCHK(sl->rows.ensure((size_t)m128 * 128 * 16));
    unsigned gx = ot_cols_groups(m128);
    hipLaunchKernelGGL((ot_cols_kernel<0>), dim3(gx, 128), dim3(1024), 0, r->st, r->rk0, r->rk1, r->ctr, m128,
                       (const uint4 *)sl->cbits.p, (const uint4 *)0, make_uint4(0, 0, 0, 0), (uint4 *)r->T0.p, U);
    OTLAUNCH();
    if (!r->dev_io) OTCHK(hipMemcpyAsync(u_out, U, cbytes, hipMemcpyDeviceToHost, r->st));   // overlaps the transpose
    const uint64_t mp = (uint64_t)m128 * 128;
    hipLaunchKernelGGL(ot_transpose_kernel, dim3((unsigned)((mp / 512 + 4) / 4)), dim3(256), 0, r->st, (const uint64_t *)r->T0.p, m128,
                       (uint4 *)sl->rows.p, mp);
    OTLAUNCH();
    OTCHK(hipStreamSynchronize(r->st));
    r->ctr += m128;
    return LGC_OK;
}
static int send_extend(lgc_ot_sender *s, uint64_t m, const uint8_t *u_in) {
    const uint32_t m128 = (uint32_t)(round128(m) / 128);
    const size_t cbytes = (size_t)128 * m128 * 16;
    OTCHK(s->Q.ensure(cbytes));
    OTCHK(s->rows.ensure((size_t)m128 * 128 * 16));
    const uint4 *U = reinterpret_cast<const uint4 *>(u_in);
    if (!s->dev_io) {
        OTCHK(s->U.ensure(cbytes));
        OTCHK(hipMemcpyAsync(s->U.p, u_in, cbytes, hipMemcpyHostToDevice, s->st));
        U = static_cast<const uint4 *>(s->U.p);
    }
    unsigned gx = ot_cols_groups(m128);
    hipLaunchKernelGGL((ot_cols_kernel<1>), dim3(gx, 128), dim3(1024), 0, s->st, s->rk, (const uint32_t *)0, s->ctr, m128,
                       (const uint4 *)0, U, s->delta, (uint4 *)s->Q.p, (uint4 *)0);
    OTLAUNCH();
    const uint64_t mp = (uint64_t)m128 * 128;
    hipLaunchKernelGGL(ot_transpose_kernel, dim3((unsigned)((mp / 512 + 4) / 4)), dim3(256), 0, s->st, (const uint64_t *)s->Q.p, m128,
                       (uint4 *)s->rows.p, mp);
    OTLAUNCH();
    s->ctr += m128;
    return LGC_OK;
}

extern "C" size_t lgc_ot_u_bytes(uint64_t m) { return (size_t)(round128(m) / 128) * 128 * 16; }

static lgc_ot_receiver::Slot *slot_push(lgc_ot_receiver *r) {
    if (r->count >= kMaxRecvInFlight) return 0;
    lgc_ot_receiver::Slot *sl = &r->slot[(r->head + r->count) % kMaxRecvInFlight];
    r->count++;
    return sl;
}
static void slot_unpush(lgc_ot_receiver *r) { r->count--; }
static lgc_ot_receiver::Slot *slot_front(lgc_ot_receiver *r) { return r->count ? &r->slot[r->head] : 0; }
static void slot_pop(lgc_ot_receiver *r) { r->head = (r->head + 1) % kMaxRecvInFlight; r->count--; }

// ---------------------------------------------------------------- Gilboa
extern "C" int lgc_ot_gilboa_recv_start(lgc_ot_receiver *r, const uint64_t *a, size_t npairs, size_t n, int width, uint8_t *u_out) {
    if (!r || !a || !u_out) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    std::lock_guard<std::mutex> lock(r->mu);
    if (!r->st && lgc_stream_take(r->device, &r->st) != hipSuccess) return lgc_fail(LGC_EHIP, "hipStreamCreate failed");
    lgc_ot_receiver::Slot *sl = slot_push(r);
    if (!sl) return lgc_fail(LGC_ESTATE, "too many receives in flight (%zu)", r->count);
    struct Undo { lgc_ot_receiver *r; bool armed; ~Undo() { if (armed) slot_unpush(r); } } undo = {r, true};
    OTCHK(hipSetDevice(r->device));
    const uint64_t nw = (uint64_t)npairs * n, m = nw * (uint64_t)width;
    const uint64_t m128 = round128(m) / 128;
    const uint64_t *da = a;
    if (!r->dev_io) {
        OTCHK(sl->avals.ensure(nw * 8));
        OTCHK(hipMemcpyAsync(sl->avals.p, a, nw * 8, hipMemcpyHostToDevice, r->st));
        da = static_cast<const uint64_t *>(sl->avals.p);
    }
    sl->avals_dev = da;
    OTCHK(sl->cbits.ensure(m128 * 16));
    OTCHK(hipMemsetAsync(sl->cbits.p, 0, m128 * 16, r->st));
    hipLaunchKernelGGL(ot_pack_choice_words_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, r->st, da, nw, width,
                       (uint64_t *)sl->cbits.p);
    OTLAUNCH();
    sl->m = m; sl->npairs = npairs; sl->n = n; sl->w = width; sl->tweak_cur = r->tweak; sl->gilboa = true;
    int rc = recv_extend(r, sl, m, u_out);
    if (rc) return rc;
    r->tweak += m;
    undo.armed = false;
    return LGC_OK;
}
extern "C" int lgc_ot_gilboa_send(lgc_ot_sender *s, const uint64_t *b, size_t npairs, size_t n, int width, const uint8_t *u_in,
                                  uint64_t *y_out, uint64_t *shares) {
    if (!s || !b || !u_in || !y_out || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    OTCHK(hipSetDevice(s->device));
    if (!s->st && lgc_stream_take(s->device, &s->st) != hipSuccess) return lgc_fail(LGC_EHIP, "hipStreamCreate failed");
    const uint64_t nw = (uint64_t)npairs * n, m = nw * (uint64_t)width, mpp = (uint64_t)n * width;
    int rc = send_extend(s, m, u_in);
    if (rc) return rc;
    const uint64_t *db = b;
    uint64_t *dy = y_out;
    if (!s->dev_io) {
        OTCHK(s->b.ensure(nw * 8)); OTCHK(s->y.ensure(m * 8));
        OTCHK(hipMemcpyAsync(s->b.p, b, nw * 8, hipMemcpyHostToDevice, s->st));
        db = static_cast<const uint64_t *>(s->b.p); dy = static_cast<uint64_t *>(s->y.p);
    }
    OTCHK(s->sh.ensure(npairs * 8));
    OTCHK(hipMemsetAsync(s->sh.p, 0, npairs * 8, s->st));
    unsigned gx, gy;
    ot_pair_grid(mpp, npairs, gx, gy);
    hipLaunchKernelGGL(ot_gilboa_send_kernel, dim3(gx, gy), dim3(1024), 0, s->st, (const uint4 *)s->rows.p, s->delta, db, (uint64_t)n,
                       width, mpp, (uint64_t)npairs, s->tweak, dy, (uint64_t *)s->sh.p);
    OTLAUNCH();
    if (!s->dev_io) OTCHK(hipMemcpyAsync(y_out, dy, m * 8, hipMemcpyDeviceToHost, s->st));
    OTCHK(hipMemcpyAsync(shares, s->sh.p, npairs * 8, s->dev_io ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->st));
    OTCHK(hipStreamSynchronize(s->st));
    if (width == 32 && !s->dev_io) for (size_t q = 0; q < npairs; q++) shares[q] &= 0xffffffffull;
    s->tweak += m;
    return LGC_OK;
}
extern "C" int lgc_ot_gilboa_recv_finish(lgc_ot_receiver *r, const uint64_t *y_in, uint64_t *shares) {
    if (!r || !y_in || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    lgc_ot_receiver::Slot *sl = slot_front(r);
    if (!sl || !sl->gilboa) return lgc_fail(LGC_ESTATE, "no Gilboa receive in flight");
    struct Pop { lgc_ot_receiver *r; ~Pop() { slot_pop(r); } } pop = {r};   // the transfer is consumed on every path
    OTCHK(hipSetDevice(r->device));
    const uint64_t *dy = y_in;
    if (!r->dev_io) {
        OTCHK(r->y.ensure(sl->m * 8));
        OTCHK(hipMemcpyAsync(r->y.p, y_in, sl->m * 8, hipMemcpyHostToDevice, r->st));
        dy = static_cast<const uint64_t *>(r->y.p);
    }
    OTCHK(r->sh.ensure(sl->npairs * 8));
    OTCHK(hipMemsetAsync(r->sh.p, 0, sl->npairs * 8, r->st));
    const uint64_t mpp = sl->n * (uint64_t)sl->w;
    unsigned gx, gy;
    ot_pair_grid(mpp, sl->npairs, gx, gy);
    hipLaunchKernelGGL(ot_gilboa_recv_kernel, dim3(gx, gy), dim3(1024), 0, r->st, (const uint4 *)sl->rows.p, sl->avals_dev, sl->n, sl->w,
                       mpp, sl->npairs, sl->tweak_cur, dy, (uint64_t *)r->sh.p);
    OTLAUNCH();
    OTCHK(hipMemcpyAsync(shares, r->sh.p, sl->npairs * 8, r->dev_io ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, r->st));
    OTCHK(hipStreamSynchronize(r->st));
    if (sl->w == 32 && !r->dev_io) for (size_t q = 0; q < sl->npairs; q++) shares[q] &= 0xffffffffull;
    return LGC_OK;
}

// ---------------------------------------------------------------- labels
// choice: m bytes, one per OT (the reference's bool* sel, src/input.c:40-44)
extern "C" int lgc_ot_labels_recv_start(lgc_ot_receiver *r, const uint8_t *choice, size_t m, uint8_t *u_out) {
    if (!r || !choice || !u_out) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    lgc_ot_receiver::Slot *sl = slot_push(r);
    if (!sl) return lgc_fail(LGC_ESTATE, "too many receives in flight (%zu)", r->count);
    struct Undo { lgc_ot_receiver *r; bool armed; ~Undo() { if (armed) slot_unpush(r); } } undo = {r, true};
    OTCHK(hipSetDevice(r->device));
    const uint64_t m128 = round128(m) / 128;
    const uint8_t *dc = choice;
    if (!r->dev_io) {
        OTCHK(r->choice.ensure(m));
        OTCHK(hipMemcpyAsync(r->choice.p, choice, m, hipMemcpyHostToDevice, r->st));
        dc = static_cast<const uint8_t *>(r->choice.p);
    }
    OTCHK(sl->cbits.ensure(m128 * 16));
    const uint64_t nwords = m128 * 2;
    hipLaunchKernelGGL(ot_pack_choice_bytes_kernel, dim3((unsigned)((nwords * 64 + 255) / 256)), dim3(256), 0, r->st, dc, (uint64_t)m,
                       (uint64_t *)sl->cbits.p, nwords);
    OTLAUNCH();
    sl->m = m; sl->npairs = 0; sl->n = 0; sl->w = 0; sl->tweak_cur = r->tweak; sl->gilboa = false; sl->avals_dev = 0;
    int rc = recv_extend(r, sl, m, u_out);
    if (rc) return rc;
    r->tweak += m;
    undo.armed = false;
    return LGC_OK;
}
extern "C" int lgc_ot_labels_send(lgc_ot_sender *s, const uint8_t *msg0, const uint8_t *msg1, size_t m, const uint8_t *u_in, uint8_t *e_out) {
    if (!s || !msg0 || !msg1 || !u_in || !e_out) return lgc_fail(LGC_EINVAL, "null argument");
    OTCHK(hipSetDevice(s->device));
    int rc = send_extend(s, m, u_in);
    if (rc) return rc;
    const uint4 *d0 = reinterpret_cast<const uint4 *>(msg0), *d1 = reinterpret_cast<const uint4 *>(msg1);
    uint4 *de = reinterpret_cast<uint4 *>(e_out);
    if (!s->dev_io) {
        OTCHK(s->m0.ensure(m * 16)); OTCHK(s->m1.ensure(m * 16)); OTCHK(s->e.ensure(m * 32));
        OTCHK(hipMemcpyAsync(s->m0.p, msg0, m * 16, hipMemcpyHostToDevice, s->st));
        OTCHK(hipMemcpyAsync(s->m1.p, msg1, m * 16, hipMemcpyHostToDevice, s->st));
        d0 = (const uint4 *)s->m0.p; d1 = (const uint4 *)s->m1.p; de = (uint4 *)s->e.p;
    }
    unsigned gx = (unsigned)((m + 2047) / 2048); if (gx > kOtGroups) gx = kOtGroups;
    hipLaunchKernelGGL(ot_labels_send_kernel, dim3(gx), dim3(1024), 0, s->st, (const uint4 *)s->rows.p, s->delta, d0, d1, (uint64_t)m,
                       s->tweak, de);
    OTLAUNCH();
    if (!s->dev_io) OTCHK(hipMemcpyAsync(e_out, de, m * 32, hipMemcpyDeviceToHost, s->st));
    OTCHK(hipStreamSynchronize(s->st));
    s->tweak += m;
    return LGC_OK;
}
extern "C" int lgc_ot_labels_recv_finish(lgc_ot_receiver *r, const uint8_t *e_in, uint8_t *out) {
    if (!r || !e_in || !out) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    lgc_ot_receiver::Slot *sl = slot_front(r);
    if (!sl || sl->gilboa) return lgc_fail(LGC_ESTATE, "no label receive in flight");
    struct Pop { lgc_ot_receiver *r; ~Pop() { slot_pop(r); } } pop = {r};
    OTCHK(hipSetDevice(r->device));
    const uint4 *de = reinterpret_cast<const uint4 *>(e_in);
    uint4 *dout = reinterpret_cast<uint4 *>(out);
    if (!r->dev_io) {
        OTCHK(r->e.ensure(sl->m * 32)); OTCHK(r->out.ensure(sl->m * 16));
        OTCHK(hipMemcpyAsync(r->e.p, e_in, sl->m * 32, hipMemcpyHostToDevice, r->st));
        de = (const uint4 *)r->e.p; dout = (uint4 *)r->out.p;
    }
    unsigned gx = (unsigned)((sl->m + 2047) / 2048); if (gx > kOtGroups) gx = kOtGroups;
    hipLaunchKernelGGL(ot_labels_recv_kernel, dim3(gx), dim3(1024), 0, r->st, (const uint4 *)sl->rows.p, (const uint64_t *)sl->cbits.p, de,
                       sl->m, sl->tweak_cur, dout);
    OTLAUNCH();
    if (!r->dev_io) OTCHK(hipMemcpyAsync(out, dout, sl->m * 16, hipMemcpyDeviceToHost, r->st));
    OTCHK(hipStreamSynchronize(r->st));
    return LGC_OK;
}
