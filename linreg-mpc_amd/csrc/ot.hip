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

#include <deque>
#include <mutex>
#include <vector>

#include "../../include/linreg_gc.h"
#include "hip_scope.h"
#include "gc_device.h"

using namespace gc;

int lgc_fail(int code, const char *fmt, ...);
int lgc_need_device(int device);
int lgc_upload_constants();

#define OTCHK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) return lgc_fail(LGC_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- column PRG.  grid.y = column j (0..127); each lane produces AES_kj(ctr0 + b) for block b.
// MODE 0 (receiver): T0[j][b] = G(k0), U[j][b] = G(k0) ^ G(k1) ^ c[b]
// MODE 1 (sender)  : Q[j][b]  = G(k)  ^ (delta_j ? U[j][b] : 0)
template <int MODE>
__global__ void __launch_bounds__(1024)
ot_cols_kernel(const uint32_t *rk0, const uint32_t *rk1, uint64_t ctr0, uint32_t m128, const uint4 *cbits,
               const uint4 *Uin, uint4 delta, uint4 *out0, uint4 *out1) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    __shared__ uint32_t sk0[44], sk1[44];
    lds_tab_fill(lds_te0);
    const uint32_t j = blockIdx.y;
    if (threadIdx.x < 44) {
        sk0[threadIdx.x] = rk0[j * 44 + threadIdx.x];
        if (MODE == 0) sk1[threadIdx.x] = rk1[j * 44 + threadIdx.x];
    }
    __syncthreads();
    LdsTab lt = lds_tab_make(lds_te0);
    const uint32_t dw = j < 32 ? delta.x : (j < 64 ? delta.y : (j < 96 ? delta.z : delta.w));
    const bool dj = (dw >> (j & 31)) & 1u;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < m128; b += gridDim.x * blockDim.x) {
        const uint64_t c = ctr0 + b;
        if (MODE == 0) {
            uint32_t s0[1][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}};
            uint32_t s1[1][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}};
            aes_encrypt_n<1, LdsTab>(lt, sk0, s0);
            aes_encrypt_n<1, LdsTab>(lt, sk1, s1);
            uint4 cb = cbits[b];
            out0[(size_t)j * m128 + b] = make_uint4(s0[0][0], s0[0][1], s0[0][2], s0[0][3]);
            out1[(size_t)j * m128 + b] = make_uint4(s0[0][0] ^ s1[0][0] ^ cb.x, s0[0][1] ^ s1[0][1] ^ cb.y,
                                                    s0[0][2] ^ s1[0][2] ^ cb.z, s0[0][3] ^ s1[0][3] ^ cb.w);
        } else {
            uint32_t s0[1][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}};
            aes_encrypt_n<1, LdsTab>(lt, sk0, s0);
            uint4 u = Uin[(size_t)j * m128 + b];
            uint32_t k = dj ? 0xffffffffu : 0u;
            out0[(size_t)j * m128 + b] = make_uint4(s0[0][0] ^ (u.x & k), s0[0][1] ^ (u.y & k), s0[0][2] ^ (u.z & k),
                                                    s0[0][3] ^ (u.w & k));
        }
    }
}

// ---- 128 x m bit transpose: cols[j][m128] (bit i of column j = bit (i & 127) of block i >> 7)
// -> rows[i] = 128 bits (bit j = column j).  One wave per 64 consecutive OTs.
__global__ void __launch_bounds__(256)
ot_transpose_kernel(const uint64_t *cols, uint32_t m128, uint4 *rows, uint64_t m) {
    const int lane = threadIdx.x & 63;
    const uint64_t wv = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t i0 = wv * 64;
    if (i0 >= m) return;
    uint64_t r0 = 0, r1 = 0;   // row bits for columns 0..63 and 64..127
#pragma unroll
    for (int half = 0; half < 2; half++) {
        // lane l holds bits i0..i0+63 of column (64 * half + l): one 64-bit word of that column
        const uint64_t w = cols[((size_t)(64 * half + lane) * m128) * 2 + (i0 >> 6)];
        uint64_t acc = 0;
        for (int t = 0; t < 64; t++) {
            uint64_t mk = __ballot((w >> t) & 1ull);
            if (lane == t) acc = mk;
        }
        if (half == 0) r0 = acc; else r1 = acc;
    }
    if (i0 + lane < m) rows[i0 + lane] = make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32));
}

__device__ __forceinline__ Lbl u4_lbl(uint4 v) { Lbl l = {v.x, v.y, v.z, v.w}; return l; }

// ---- Gilboa sender: OT i = (q * n + k) * w + bit.  y_i and per-pair share -sum x0.
__global__ void __launch_bounds__(1024)
ot_gilboa_send_kernel(const uint4 *rows, uint4 delta, const uint64_t *bvals, uint64_t n, int w, uint64_t m_per_pair,
                      uint64_t tweak0, uint64_t *y, uint64_t *shares) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    lds_tab_fill(lds_te0);
    LdsTab lt = lds_tab_make(lds_te0);
    const uint32_t q = blockIdx.y;
    const uint64_t mask = w == 32 ? 0xffffffffull : ~0ull;
    uint64_t acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m_per_pair; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (uint64_t)q * m_per_pair + t;
        const uint64_t k = t / (uint64_t)w;
        const int bit = (int)(t % (uint64_t)w);
        Lbl x[2] = {u4_lbl(rows[i]), lxor(u4_lbl(rows[i]), u4_lbl(delta))};
        uint64_t tw[2] = {tweak0 + i, tweak0 + i};
        Lbl h[2];
        hash_n<2, LdsTab>(lt, c_rk, x, tw, h);
        uint64_t x0 = ((uint64_t)h[0].x | ((uint64_t)h[0].y << 32)) & mask;
        uint64_t h1 = ((uint64_t)h[1].x | ((uint64_t)h[1].y << 32)) & mask;
        uint64_t d = (bvals[(uint64_t)q * n + k] << bit) & mask;
        y[i] = (x0 + d - h1) & mask;
        acc -= x0;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&shares[q], (unsigned long long)acc);
}

__global__ void __launch_bounds__(1024)
ot_gilboa_recv_kernel(const uint4 *rows, const uint64_t *avals, uint64_t n, int w, uint64_t m_per_pair, uint64_t tweak0,
                      const uint64_t *y, uint64_t *shares) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    lds_tab_fill(lds_te0);
    LdsTab lt = lds_tab_make(lds_te0);
    const uint32_t q = blockIdx.y;
    const uint64_t mask = w == 32 ? 0xffffffffull : ~0ull;
    uint64_t acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m_per_pair; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (uint64_t)q * m_per_pair + t;
        const uint64_t k = t / (uint64_t)w;
        const int bit = (int)(t % (uint64_t)w);
        Lbl x = u4_lbl(rows[i]);
        uint64_t tw = tweak0 + i;
        Lbl h;
        hash_n<1, LdsTab>(lt, c_rk, &x, &tw, &h);
        uint64_t v = ((uint64_t)h.x | ((uint64_t)h.y << 32)) & mask;
        if ((avals[(uint64_t)q * n + k] >> bit) & 1ull) v += y[i];
        acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&shares[q], (unsigned long long)acc);
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

// ---- 1-of-2 OT of 16-byte messages
__global__ void __launch_bounds__(1024)
ot_labels_send_kernel(const uint4 *rows, uint4 delta, const uint4 *m0, const uint4 *m1, uint64_t m, uint64_t tweak0, uint4 *e) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    lds_tab_fill(lds_te0);
    LdsTab lt = lds_tab_make(lds_te0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        Lbl x[2] = {u4_lbl(rows[i]), lxor(u4_lbl(rows[i]), u4_lbl(delta))};
        uint64_t tw[2] = {tweak0 + i, tweak0 + i};
        Lbl h[2];
        hash_n<2, LdsTab>(lt, c_rk, x, tw, h);
        uint4 a = m0[i], b = m1[i];
        e[2 * i] = make_uint4(a.x ^ h[0].x, a.y ^ h[0].y, a.z ^ h[0].z, a.w ^ h[0].w);
        e[2 * i + 1] = make_uint4(b.x ^ h[1].x, b.y ^ h[1].y, b.z ^ h[1].z, b.w ^ h[1].w);
    }
}
__global__ void __launch_bounds__(1024)
ot_labels_recv_kernel(const uint4 *rows, const uint64_t *cbits, const uint4 *e, uint64_t m, uint64_t tweak0, uint4 *out) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    lds_tab_fill(lds_te0);
    LdsTab lt = lds_tab_make(lds_te0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        Lbl x = u4_lbl(rows[i]);
        uint64_t tw = tweak0 + i;
        Lbl h;
        hash_n<1, LdsTab>(lt, c_rk, &x, &tw, &h);
        uint32_t c = (uint32_t)(cbits[i >> 6] >> (i & 63)) & 1u;
        uint4 ev = e[2 * i + c];
        out[i] = make_uint4(ev.x ^ h.x, ev.y ^ h.y, ev.z ^ h.z, ev.w ^ h.w);
    }
}

// =============================================================== sessions
struct lgc_ot_sender {
    int device;
    uint4 delta;
    uint32_t *rk;          // 128 x 44 round keys of G(k_j^{Delta_j})
    uint64_t ctr;          // PRG stream position (blocks), same on both sides
    uint64_t tweak;        // hash tweak counter (OT index), same on both sides
};
struct lgc_ot_receiver {
    int device;
    uint32_t *rk0, *rk1;
    uint64_t ctr, tweak;
    // state between *_start and *_finish
    uint4 *rows;
    uint64_t *cbits;
    uint64_t *avals;
    uint64_t m, npairs, n;
    int w;
    uint64_t tweak_cur;
    // receives started but not finished yet, oldest first: *_recv_start parks its state here and
    // *_recv_finish takes the oldest, so a caller may keep a few batches in flight (one thread
    // starting and sending u, another receiving the replies and finishing)
    struct Pending { uint4 *rows; uint64_t *cbits, *avals; uint64_t m, npairs, n; int w; uint64_t tweak_cur; };
    std::deque<Pending> fifo;
    std::mutex mu;
};
static const size_t kMaxRecvInFlight = 4;
static void recv_park(lgc_ot_receiver *r) {
    lgc_ot_receiver::Pending p = {r->rows, r->cbits, r->avals, r->m, r->npairs, r->n, r->w, r->tweak_cur};
    r->fifo.push_back(p);
    r->rows = 0; r->cbits = 0; r->avals = 0; r->m = 0;
}
static bool recv_unpark(lgc_ot_receiver *r) {
    if (r->fifo.empty()) return false;
    lgc_ot_receiver::Pending p = r->fifo.front();
    r->fifo.pop_front();
    r->rows = p.rows; r->cbits = p.cbits; r->avals = p.avals; r->m = p.m; r->npairs = p.npairs; r->n = p.n; r->w = p.w;
    r->tweak_cur = p.tweak_cur;
    return true;
}

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

extern "C" int lgc_ot_sender_create(lgc_ot_sender **out, int device, const uint8_t delta[16], const uint8_t seeds[128][16]) {
    if (!out || !delta || !seeds) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    rc = lgc_upload_constants();
    if (rc) return rc;
    lgc_ot_sender *s = new lgc_ot_sender();
    s->device = device; s->ctr = 0; s->tweak = 0; s->rk = 0;
    memcpy(&s->delta, delta, 16);
    rc = upload_keys(seeds, &s->rk);
    if (rc) { delete s; return rc; }
    *out = s;
    return LGC_OK;
}
extern "C" void lgc_ot_sender_destroy(lgc_ot_sender *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->rk) (void)hipFree(s->rk);
    delete s;
}
extern "C" int lgc_ot_receiver_create(lgc_ot_receiver **out, int device, const uint8_t seeds0[128][16], const uint8_t seeds1[128][16]) {
    if (!out || !seeds0 || !seeds1) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    rc = lgc_upload_constants();
    if (rc) return rc;
    lgc_ot_receiver *r = new lgc_ot_receiver();   // value-initialised: scalars and pointers are zero
    r->device = device;
    rc = upload_keys(seeds0, &r->rk0);
    if (!rc) rc = upload_keys(seeds1, &r->rk1);
    if (rc) { delete r; return rc; }
    *out = r;
    return LGC_OK;
}
static void recv_drop_state(lgc_ot_receiver *r) {
    if (r->rows) (void)hipFree(r->rows);
    if (r->cbits) (void)hipFree(r->cbits);
    if (r->avals) (void)hipFree(r->avals);
    r->rows = 0; r->cbits = 0; r->avals = 0; r->m = 0;
}
extern "C" void lgc_ot_receiver_destroy(lgc_ot_receiver *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    recv_drop_state(r);
    while (recv_unpark(r)) recv_drop_state(r);
    if (r->rk0) (void)hipFree(r->rk0);
    if (r->rk1) (void)hipFree(r->rk1);
    delete r;
}

static inline uint64_t round128(uint64_t m) { return (m + 127) / 128 * 128; }

// receiver: columns + u + transpose for choice bits already on the device (cbits: padded to m128 blocks)
static int recv_extend(lgc_ot_receiver *r, uint64_t m, uint8_t *u_out) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    const uint32_t m128 = (uint32_t)(round128(m) / 128);
    uint4 *T0 = 0, *U = 0;
    OTCHK(hipMalloc(&T0, (size_t)128 * m128 * 16)); dev_guard.add(T0);
    OTCHK(hipMalloc(&U, (size_t)128 * m128 * 16)); dev_guard.add(U);
    OTCHK(hipMalloc(&r->rows, (size_t)m128 * 128 * 16));
    unsigned gx = (m128 + 1023) / 1024; if (gx > 64) gx = 64;
    hipLaunchKernelGGL((ot_cols_kernel<0>), dim3(gx, 128), dim3(1024), 0, 0, r->rk0, r->rk1, r->ctr, m128,
                       (const uint4 *)r->cbits, (const uint4 *)0, make_uint4(0, 0, 0, 0), T0, U);
    const uint64_t mp = (uint64_t)m128 * 128;
    hipLaunchKernelGGL(ot_transpose_kernel, dim3((unsigned)((mp / 64 + 3) / 4)), dim3(256), 0, 0, (const uint64_t *)T0, m128,
                       r->rows, mp);
    OTCHK(hipMemcpy(u_out, U, (size_t)128 * m128 * 16, hipMemcpyDeviceToHost));

    r->ctr += m128;
    return LGC_OK;
}
static int send_extend(lgc_ot_sender *s, uint64_t m, const uint8_t *u_in, uint4 **rows_out) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    const uint32_t m128 = (uint32_t)(round128(m) / 128);
    uint4 *Q = 0, *U = 0, *rows = 0;
    OTCHK(hipMalloc(&Q, (size_t)128 * m128 * 16)); dev_guard.add(Q);
    OTCHK(hipMalloc(&U, (size_t)128 * m128 * 16)); dev_guard.add(U);
    OTCHK(hipMalloc(&rows, (size_t)m128 * 128 * 16)); dev_guard.add(rows);
    OTCHK(hipMemcpy(U, u_in, (size_t)128 * m128 * 16, hipMemcpyHostToDevice));
    unsigned gx = (m128 + 1023) / 1024; if (gx > 64) gx = 64;
    hipLaunchKernelGGL((ot_cols_kernel<1>), dim3(gx, 128), dim3(1024), 0, 0, s->rk, (const uint32_t *)0, s->ctr, m128,
                       (const uint4 *)0, (const uint4 *)U, s->delta, Q, (uint4 *)0);
    const uint64_t mp = (uint64_t)m128 * 128;
    hipLaunchKernelGGL(ot_transpose_kernel, dim3((unsigned)((mp / 64 + 3) / 4)), dim3(256), 0, 0, (const uint64_t *)Q, m128,
                       rows, mp);

    s->ctr += m128;
    dev_guard.release(rows);   // ownership passes to the caller
    *rows_out = rows;
    return LGC_OK;
}

extern "C" size_t lgc_ot_u_bytes(uint64_t m) { return (size_t)(round128(m) / 128) * 128 * 16; }

// ---------------------------------------------------------------- Gilboa
extern "C" int lgc_ot_gilboa_recv_start(lgc_ot_receiver *r, const uint64_t *a, size_t npairs, size_t n, int width, uint8_t *u_out) {
    if (!r || !a || !u_out) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->fifo.size() >= kMaxRecvInFlight) return lgc_fail(LGC_ESTATE, "too many receives in flight (%zu)", r->fifo.size());
    OTCHK(hipSetDevice(r->device));
    const uint64_t nw = (uint64_t)npairs * n, m = nw * (uint64_t)width;
    const uint64_t m128 = round128(m) / 128;
    OTCHK(hipMalloc(&r->avals, nw * 8));
    OTCHK(hipMemcpy(r->avals, a, nw * 8, hipMemcpyHostToDevice));
    OTCHK(hipMalloc(&r->cbits, m128 * 16));
    OTCHK(hipMemset(r->cbits, 0, m128 * 16));
    hipLaunchKernelGGL(ot_pack_choice_words_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, r->avals, nw, width, r->cbits);
    r->m = m; r->npairs = npairs; r->n = n; r->w = width; r->tweak_cur = r->tweak;
    int rc = recv_extend(r, m, u_out);
    if (rc) { recv_drop_state(r); return rc; }
    r->tweak += m;
    recv_park(r);
    return LGC_OK;
}
extern "C" int lgc_ot_gilboa_send(lgc_ot_sender *s, const uint64_t *b, size_t npairs, size_t n, int width, const uint8_t *u_in,
                                  uint64_t *y_out, uint64_t *shares) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!s || !b || !u_in || !y_out || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    OTCHK(hipSetDevice(s->device));
    const uint64_t nw = (uint64_t)npairs * n, m = nw * (uint64_t)width, mpp = (uint64_t)n * width;
    uint4 *rows = 0;
    int rc = send_extend(s, m, u_in, &rows);
    if (rc) return rc;
    dev_guard.add(rows);
    uint64_t *db = 0, *dy = 0, *dsh = 0;
    OTCHK(hipMalloc(&db, nw * 8)); dev_guard.add(db); OTCHK(hipMalloc(&dy, m * 8)); dev_guard.add(dy); OTCHK(hipMalloc(&dsh, npairs * 8)); dev_guard.add(dsh);
    OTCHK(hipMemcpy(db, b, nw * 8, hipMemcpyHostToDevice));
    OTCHK(hipMemset(dsh, 0, npairs * 8));
    unsigned gx = (unsigned)((mpp + 1023) / 1024); if (gx > 256) gx = 256;
    hipLaunchKernelGGL(ot_gilboa_send_kernel, dim3(gx, (unsigned)npairs), dim3(1024), 0, 0, rows, s->delta, db, (uint64_t)n, width,
                       mpp, s->tweak, dy, dsh);
    OTCHK(hipMemcpy(y_out, dy, m * 8, hipMemcpyDeviceToHost));
    OTCHK(hipMemcpy(shares, dsh, npairs * 8, hipMemcpyDeviceToHost));
    if (width == 32) for (size_t q = 0; q < npairs; q++) shares[q] &= 0xffffffffull;

    s->tweak += m;
    return LGC_OK;
}
extern "C" int lgc_ot_gilboa_recv_finish(lgc_ot_receiver *r, const uint64_t *y_in, uint64_t *shares) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!r || !y_in || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->fifo.empty() || !r->fifo.front().avals) return lgc_fail(LGC_ESTATE, "no Gilboa receive in flight");
    recv_unpark(r);
    OTCHK(hipSetDevice(r->device));
    uint64_t *dy = 0, *dsh = 0;
    OTCHK(hipMalloc(&dy, r->m * 8)); dev_guard.add(dy); OTCHK(hipMalloc(&dsh, r->npairs * 8)); dev_guard.add(dsh);
    OTCHK(hipMemcpy(dy, y_in, r->m * 8, hipMemcpyHostToDevice));
    OTCHK(hipMemset(dsh, 0, r->npairs * 8));
    const uint64_t mpp = r->n * (uint64_t)r->w;
    unsigned gx = (unsigned)((mpp + 1023) / 1024); if (gx > 256) gx = 256;
    hipLaunchKernelGGL(ot_gilboa_recv_kernel, dim3(gx, (unsigned)r->npairs), dim3(1024), 0, 0, r->rows, r->avals, r->n, r->w, mpp,
                       r->tweak_cur, dy, dsh);
    OTCHK(hipMemcpy(shares, dsh, r->npairs * 8, hipMemcpyDeviceToHost));
    if (r->w == 32) for (size_t q = 0; q < r->npairs; q++) shares[q] &= 0xffffffffull;

    recv_drop_state(r);
    return LGC_OK;
}

// ---------------------------------------------------------------- labels
// choice: m bytes, one per OT (the reference's bool* sel, src/input.c:40-44)
extern "C" int lgc_ot_labels_recv_start(lgc_ot_receiver *r, const uint8_t *choice, size_t m, uint8_t *u_out) {
    if (!r || !choice || !u_out) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->fifo.size() >= kMaxRecvInFlight) return lgc_fail(LGC_ESTATE, "too many receives in flight (%zu)", r->fifo.size());
    OTCHK(hipSetDevice(r->device));
    const uint64_t m128 = round128(m) / 128;
    std::vector<uint64_t> packed(m128 * 2, 0);
    for (size_t i = 0; i < m; i++) if (choice[i]) packed[i >> 6] |= 1ull << (i & 63);
    OTCHK(hipMalloc(&r->cbits, m128 * 16));
    OTCHK(hipMemcpy(r->cbits, packed.data(), m128 * 16, hipMemcpyHostToDevice));
    r->m = m; r->npairs = 0; r->n = 0; r->w = 0; r->tweak_cur = r->tweak;
    int rc = recv_extend(r, m, u_out);
    if (rc) { recv_drop_state(r); return rc; }
    r->tweak += m;
    recv_park(r);
    return LGC_OK;
}
extern "C" int lgc_ot_labels_send(lgc_ot_sender *s, const uint8_t *msg0, const uint8_t *msg1, size_t m, const uint8_t *u_in, uint8_t *e_out) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!s || !msg0 || !msg1 || !u_in || !e_out) return lgc_fail(LGC_EINVAL, "null argument");
    OTCHK(hipSetDevice(s->device));
    uint4 *rows = 0;
    int rc = send_extend(s, m, u_in, &rows);
    if (rc) return rc;
    dev_guard.add(rows);
    uint4 *d0 = 0, *d1 = 0, *de = 0;
    OTCHK(hipMalloc(&d0, m * 16)); dev_guard.add(d0); OTCHK(hipMalloc(&d1, m * 16)); dev_guard.add(d1); OTCHK(hipMalloc(&de, m * 32)); dev_guard.add(de);
    OTCHK(hipMemcpy(d0, msg0, m * 16, hipMemcpyHostToDevice));
    OTCHK(hipMemcpy(d1, msg1, m * 16, hipMemcpyHostToDevice));
    unsigned gx = (unsigned)((m + 1023) / 1024); if (gx > 512) gx = 512;
    hipLaunchKernelGGL(ot_labels_send_kernel, dim3(gx), dim3(1024), 0, 0, rows, s->delta, d0, d1, (uint64_t)m, s->tweak, de);
    OTCHK(hipMemcpy(e_out, de, m * 32, hipMemcpyDeviceToHost));

    s->tweak += m;
    return LGC_OK;
}
extern "C" int lgc_ot_labels_recv_finish(lgc_ot_receiver *r, const uint8_t *e_in, uint8_t *out) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!r || !e_in || !out) return lgc_fail(LGC_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->fifo.empty() || r->fifo.front().avals) return lgc_fail(LGC_ESTATE, "no label receive in flight");
    recv_unpark(r);
    OTCHK(hipSetDevice(r->device));
    uint4 *de = 0, *dout = 0;
    OTCHK(hipMalloc(&de, r->m * 32)); dev_guard.add(de); OTCHK(hipMalloc(&dout, r->m * 16)); dev_guard.add(dout);
    OTCHK(hipMemcpy(de, e_in, r->m * 32, hipMemcpyHostToDevice));
    unsigned gx = (unsigned)((r->m + 1023) / 1024); if (gx > 512) gx = 512;
    hipLaunchKernelGGL(ot_labels_recv_kernel, dim3(gx), dim3(1024), 0, 0, r->rows, r->cbits, de, r->m, r->tweak_cur, dout);
    OTCHK(hipMemcpy(out, dout, r->m * 16, hipMemcpyDeviceToHost));

    recv_drop_state(r);
    return LGC_OK;
}
