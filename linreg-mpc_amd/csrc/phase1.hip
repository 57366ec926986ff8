// phase1.hip -- phase-1 aggregation on the MI355X: additive shares of X^T X and
// X^T y in the ring Z_2^w (w = 32 or 64).
//
// Replaces the arithmetic of the reference's src/phase1.c:
//   inner_product_local (14-20)            -> p1_gram_kernel  (wrap-around u64 Gram, LDS tiled)
//   diagonal special case (562-567, 364-369)-> p1_diag_kernel  (IEEE double, k ascending, no FMA)
//   run_trusted_initializer PRG + <x,y> (241-287) -> ti_prg_kernel (AES-128-CTR) + p1_dot_kernel
//   inner_product_ti masking / shares (148-236)   -> p1_mask_kernel, p1_dot_kernel
// The transport (TCP mesh, length-prefixed protobuf messages, phase1.c:100-145) stays on
// the host and is out of scope here; these entry points take and return host buffers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"
#include "hip_scope.h"
#include "gc_device.h"

using namespace gc;

int lgc_fail(int code, const char *fmt, ...);
int lgc_need_device(int device);
__global__ void p1_tu_touch_kernel() {}
hipError_t p1_tu_touch(hipStream_t st) { hipLaunchKernelGGL(p1_tu_touch_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }

#define P1CHK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) return lgc_fail(LGC_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct lgc_p1 {
    int device, w, p;
    size_t n, d;
    int64_t *X;      // n x (d + 1) row-major, column d = y (zero when this party does not own y)
    bool have_y;
    bool dev_io;     // lgc_p1_set_device_io: the vector arguments of mask / dot / ti_a_batch are device memory
};

// ---- wrap-around Gram block: C[a][b] = sum_k X[k][cols[a]] * X[k][cols[b]]  (mod 2^64)
// 64 x 64 output tile per workgroup, 4 x 4 per thread, K staged through LDS in slabs of 16 rows;
// split-K partial sums are combined with integer atomics (exact and order-independent).
#define P1_KT 16
__global__ void __launch_bounds__(256)
p1_gram_kernel(const int64_t *X, size_t n, size_t ld, const uint32_t *cols, uint32_t L, uint64_t *C, size_t kchunk) {
    __shared__ uint64_t As[P1_KT][64], Bs[P1_KT][64];
    const uint32_t i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    if (j0 > i0) return;   // lower triangle of tiles only
    const size_t k0 = (size_t)blockIdx.z * kchunk;
    const size_t k1 = k0 + kchunk < n ? k0 + kchunk : n;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    uint64_t acc[4][4] = {};
    const int lc = threadIdx.x & 63, lr = threadIdx.x >> 6;   // loader: 4 rows x 64 columns per pass
    const uint32_t ca = i0 + lc < L ? cols[i0 + lc] : 0xffffffffu;
    const uint32_t cb = j0 + lc < L ? cols[j0 + lc] : 0xffffffffu;
    for (size_t kb = k0; kb < k1; kb += P1_KT) {
#pragma unroll
        for (int r = 0; r < P1_KT; r += 4) {
            size_t k = kb + r + lr;
            As[r + lr][lc] = (k < k1 && ca != 0xffffffffu) ? (uint64_t)X[k * ld + ca] : 0;
            Bs[r + lr][lc] = (k < k1 && cb != 0xffffffffu) ? (uint64_t)X[k * ld + cb] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < P1_KT; kk++) {
            uint64_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = As[kk][ty * 4 + u]; b[u] = Bs[kk][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[u][v] += a[u] * b[v];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            uint32_t i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
            if (i < L && j < L && j <= i) atomicAdd((unsigned long long *)&C[(size_t)i * L + j], (unsigned long long)acc[u][v]);
        }
}

// ---- diagonal: xy += pow(fixed_to_double(x_k, p), 2) * pow(2, p), k ascending, in IEEE double;
// share = double_to_fixed(xy / d, p)  (src/phase1.c:562-567).  Explicit round-to-nearest
// multiplies/adds so that no FMA contraction can change the rounding.  Only the additions are
// order-dependent: a workgroup owns 16 columns; 15 loader waves compute the terms of a 600-row
// chunk in parallel into LDS while wave 0 adds the previous chunk's terms in k order (the bound
// is n dependent double additions per column).
#define P1_DC 16        /* columns per workgroup */
#define P1_DR 600       /* rows per chunk: 15 waves x 4 row phases x 10 rows */
__global__ void __launch_bounds__(1024)
p1_diag_kernel(const int64_t *X, size_t n, size_t ld, const uint32_t *cols, uint32_t L,
               int p, int w, double normalizer2, uint64_t *out) {
    __shared__ double term[2][P1_DR][P1_DC];      // 150 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * P1_DC + (lane & 15);
    const bool live = t < L;
    const uint32_t c = live ? cols[t] : 0;
    const double scale = (double)(1ll << p);
    const size_t nchunks = (n + P1_DR - 1) / P1_DR;
    double xy = 0.0;
    for (size_t ch = 0; ch <= nchunks; ch++) {
        if (wave > 0 && ch < nchunks) {             // loaders: chunk ch -> term[ch & 1]
            const size_t k0 = ch * P1_DR;
            const int r0 = (wave - 1) * 4 + (lane >> 4);          // 0..59
            int64_t x[10];
#pragma unroll
            for (int u = 0; u < 10; u++) {
                size_t k = k0 + (size_t)(r0 + 60 * u);
                x[u] = (live && k < n) ? X[k * ld + c] : 0;
            }
#pragma unroll
            for (int u = 0; u < 10; u++) {
                double v = __ddiv_rn((double)x[u], scale);
                term[ch & 1][r0 + 60 * u][lane & 15] = __dmul_rn(__dmul_rn(v, v), scale);
            }
        }
        if (wave == 0 && ch > 0 && lane < P1_DC) {  // adder: chunk ch - 1, rows in k order
            const size_t k0 = (ch - 1) * P1_DR;
            const int rows = (int)((n - k0 < (size_t)P1_DR) ? n - k0 : (size_t)P1_DR);
            const double *tp = &term[(ch - 1) & 1][0][lane];
            for (int r = 0; r < rows; r++) xy = __dadd_rn(xy, tp[r * P1_DC]);
        }
        __syncthreads();
    }
    if (wave != 0 || lane >= P1_DC || !live) return;
    double tq = __dmul_rn(__ddiv_rn(xy, normalizer2), scale);
    uint64_t r;
    if (w == 32) {   // (int32_t) cast with the x86 "integer indefinite" result when out of range
        if (!(tq > -2147483649.0 && tq < 2147483648.0)) r = 0x80000000ull;
        else r = (uint64_t)(uint32_t)(int32_t)tq;
    } else {
        if (!(tq >= -9223372036854775808.0 && tq < 9223372036854775808.0)) r = 0x8000000000000000ull;
        else r = (uint64_t)(int64_t)tq;
    }
    out[t] = r;
}

// ---- batched masking: out[q][k] = X[k][col[q]] + sign * V[q][k]   (mod 2^64)
// m = 2^w - 1: the message is truncated to the protocol width HERE, so that a buffer another party maps (--ti_ring)
// never holds bits 32..63 of X +- v at w = 32 (X is stored sign-extended)
__global__ void p1_mask_kernel(const int64_t *X, size_t n, size_t ld, const uint32_t *cols, const uint64_t *V,
                               int sign, uint64_t *out, uint64_t m) {
    const uint32_t q = blockIdx.y;
    const uint32_t c = cols[q];
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)X[k * ld + c], v = V[(size_t)q * n + k];
        out[(size_t)q * n + k] = (sign > 0 ? x + v : x - v) & m;
    }
}

// ---- single-pair forms used by the per-pair TI protocol: no column-index upload, the sum lands
// next to the vector it belongs to (one copy back); up to 64 workgroups keep the strided column
// reads in flight (a single workgroup was latency-bound: config 4 went from 90 s to 105 s)
__device__ __forceinline__ uint64_t p1_block_sum(uint64_t acc) {
    __shared__ uint64_t part[16];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    uint64_t t = 0;
    if (threadIdx.x == 0) for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) t += part[i];
    return t;   // valid in thread 0
}
// out[k] = X[k][col] + sign * V[k]
__global__ void __launch_bounds__(1024)
p1_mask1_kernel(const int64_t *X, size_t n, size_t ld, uint32_t col, const uint64_t *V, int sign, uint64_t *out, uint64_t m) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)X[k * ld + col], v = V[k];
        out[k] = (sign > 0 ? x + v : x - v) & m;
    }
}
// out[0] = sum_k A[k] * (Bv ? Bv[k] : X[k][col])
__global__ void __launch_bounds__(1024)
p1_dot1_kernel(const uint64_t *A, const uint64_t *Bv, const int64_t *X, size_t ld, uint32_t col, size_t n, uint64_t *out) {
    uint64_t acc = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        acc += A[k] * (Bv ? Bv[k] : (uint64_t)X[k * ld + col]);
    uint64_t t = p1_block_sum(acc);
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&out[0], (unsigned long long)t);   // out[0] zeroed by the host
}
// party a of inner_product_ti in one pass (phase1.c:186-196): out[k] = a[k] - y[k] for k < n, and
// out[n] = sum_k in[k] * y[k]   (in = b + x from party b, y from the TI)
__global__ void __launch_bounds__(1024)
p1_ti_a_kernel(const int64_t *X, size_t n, size_t ld, uint32_t col, const uint64_t *y, const uint64_t *in, uint64_t *out, uint64_t m) {
    uint64_t acc = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        uint64_t yk = y[k];
        out[k] = ((uint64_t)X[k * ld + col] - yk) & m;
        acc += in[k] * yk;
    }
    uint64_t t = p1_block_sum(acc);
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&out[n], (unsigned long long)t);   // out[n] zeroed by the host
}

// the same for a run of pairs with one peer: grid.y strides over the pairs; acc[q] zeroed by the host
__global__ void __launch_bounds__(256)
p1_ti_a_batch_kernel(const int64_t *X, size_t n, size_t ld, const uint32_t *cols, size_t npairs, const uint64_t *y, const uint64_t *in,
                     uint64_t *out, uint64_t *acc_out, uint64_t m) {
    for (size_t q = blockIdx.y; q < npairs; q += gridDim.y) {
        const uint32_t col = cols[q];
        uint64_t acc = 0;
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
            uint64_t yk = y[q * n + k];
            out[q * n + k] = ((uint64_t)X[k * ld + col] - yk) & m;
            acc += in[q * n + k] * yk;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&acc_out[q], (unsigned long long)acc);
    }
}

// ---- batched wrap-around dot products: out[q] = sum_k A[q][k] * B[q][k]; B is either a vector
// batch (colsB == NULL) or columns of X
__global__ void __launch_bounds__(256)
p1_dot_kernel(const uint64_t *A, const uint64_t *Bv, const int64_t *X, size_t ld, const uint32_t *colsB, size_t n,
              uint64_t *out) {
    const uint32_t q = blockIdx.y;
    uint64_t acc = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        uint64_t a = A[(size_t)q * n + k];
        uint64_t b = colsB ? (uint64_t)X[k * ld + colsB[q]] : Bv[(size_t)q * n + k];
        acc += a * b;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&out[q], (unsigned long long)acc);
}

// ---- AES-128-CTR keystream: block c = AES_k(c) (little-endian 128-bit counter, nonce 0).
// rk: 44 round-key words of the session key.  One block per lane per iteration.
__global__ void __launch_bounds__(1024)
ti_prg_kernel(const uint32_t *rk, uint64_t first_block, uint64_t nblocks, uint4 *out) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    __shared__ uint32_t srk[44];
    lds_tab_fill(lds_te0);
    if (threadIdx.x < 44) srk[threadIdx.x] = rk[threadIdx.x];
    __syncthreads();
    LdsTab lt = lds_tab_make(lds_te0);
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = first_block + b;
        uint32_t s[1][4] = {{(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u}};
        aes_encrypt_n<1, LdsTab>(lt, srk, s);
        out[b] = make_uint4(s[0][0], s[0][1], s[0][2], s[0][3]);
    }
}

// split the TI keystream into x[q][n], y[q][n], r[q]: word t of pair q sits at stream word
// q * (2n + 1) + t  (w-bit little-endian words; `skip` = byte offset of the first word in ks)
__global__ void ti_unpack_kernel(const uint8_t *ks, size_t skip, size_t npairs, size_t n, int wb, uint64_t *x, uint64_t *y,
                                 uint64_t *r) {
    const size_t per = 2 * n + 1, total = npairs * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t *p = ks + skip + i * (size_t)wb;
        uint64_t v = 0;
        for (int b = 0; b < wb; b++) v |= (uint64_t)p[b] << (8 * b);
        size_t q = i / per, t = i % per;
        if (t < n) x[q * n + t] = v;
        else if (t < 2 * n) y[q * n + (t - n)] = v;
        else r[q] = v;
    }
}

// =============================================================== C ABI
extern "C" int lgc_p1_create(lgc_p1 **out, int device, size_t n, size_t d, int width, int precision) {
    if (!out) return lgc_fail(LGC_EINVAL, "null out");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    if (precision < 0 || precision >= width) return lgc_fail(LGC_EINVAL, "precision must satisfy 0 <= p < width");
    if (n < 1 || d < 1) return lgc_fail(LGC_EINVAL, "empty data");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    lgc_p1 *h = new lgc_p1();
    h->device = device; h->w = width; h->p = precision; h->n = n; h->d = d; h->X = 0; h->have_y = false; h->dev_io = false;
    hipError_t e = hipMalloc(&h->X, n * (d + 1) * sizeof(int64_t));
    if (e != hipSuccess) { delete h; return lgc_fail(LGC_ENOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    *out = h;
    return LGC_OK;
}
extern "C" void lgc_p1_destroy(lgc_p1 *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->X) (void)hipFree(h->X);
    delete h;
}
extern "C" int lgc_p1_set_data(lgc_p1 *h, const int64_t *Xq, const int64_t *yq) {
    if (!h || !Xq) return lgc_fail(LGC_EINVAL, "null argument");
    P1CHK(hipSetDevice(h->device));
    const size_t ld = h->d + 1;
    std::vector<int64_t> tmp(h->n * ld);
    for (size_t k = 0; k < h->n; k++) {
        memcpy(&tmp[k * ld], Xq + k * h->d, h->d * sizeof(int64_t));
        tmp[k * ld + h->d] = yq ? yq[k] : 0;
    }
    P1CHK(hipMemcpy(h->X, tmp.data(), tmp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    h->have_y = yq != 0;
    return LGC_OK;
}

static uint64_t maskw(int w) { return w == 32 ? 0xffffffffull : ~0ull; }

// shares of the block a data provider can compute alone (src/phase1.c:562-571; 359-384 in OT mode)
extern "C" int lgc_p1_local(lgc_p1 *h, size_t c0, size_t c1, int with_y, uint64_t *out_A, uint64_t *out_b) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!h || !out_A) return lgc_fail(LGC_EINVAL, "null argument");
    if (c0 >= c1 || c1 > h->d) return lgc_fail(LGC_EINVAL, "bad column range");
    if (with_y && (!h->have_y || !out_b)) return lgc_fail(LGC_EINVAL, "y requested but not set");
    P1CHK(hipSetDevice(h->device));
    const uint32_t own = (uint32_t)(c1 - c0), L = own + (with_y ? 1u : 0u);
    std::vector<uint32_t> cols(L);
    for (uint32_t i = 0; i < own; i++) cols[i] = (uint32_t)(c0 + i);
    if (with_y) cols[own] = (uint32_t)h->d;
    uint32_t *dcols = 0;
    uint64_t *dC = 0, *ddiag = 0;
    P1CHK(hipMalloc(&dcols, L * sizeof(uint32_t))); dev_guard.add(dcols);
    P1CHK(hipMalloc(&dC, (size_t)L * L * sizeof(uint64_t))); dev_guard.add(dC);
    P1CHK(hipMalloc(&ddiag, own * sizeof(uint64_t))); dev_guard.add(ddiag);
    P1CHK(hipMemcpy(dcols, cols.data(), L * sizeof(uint32_t), hipMemcpyHostToDevice));
    P1CHK(hipMemset(dC, 0, (size_t)L * L * sizeof(uint64_t)));
    const uint32_t tiles = (L + 63) / 64;
    // enough workgroups for 256 CUs: split K so that tiles^2/2 * ksplit >= ~1024
    size_t ksplit = 1;
    while ((size_t)tiles * tiles * ksplit < 2048 && h->n / (ksplit * 2) >= 256) ksplit *= 2;
    size_t kchunk = (h->n + ksplit - 1) / ksplit;
    kchunk = (kchunk + P1_KT - 1) / P1_KT * P1_KT;
    ksplit = (h->n + kchunk - 1) / kchunk;
    hipLaunchKernelGGL(p1_gram_kernel, dim3(tiles, tiles, (unsigned)ksplit), dim3(256), 0, 0, h->X, h->n, h->d + 1, dcols, L,
                       dC, kchunk);
    hipLaunchKernelGGL(p1_diag_kernel, dim3((own + P1_DC - 1) / P1_DC), dim3(1024), 0, 0, h->X, h->n, h->d + 1, dcols, own, h->p, h->w,
                       (double)h->d, ddiag);
    P1CHK(hipGetLastError());
    std::vector<uint64_t> C((size_t)L * L), diag(own);
    P1CHK(hipMemcpy(C.data(), dC, C.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    P1CHK(hipMemcpy(diag.data(), ddiag, own * sizeof(uint64_t), hipMemcpyDeviceToHost));

    const uint64_t m = maskw(h->w);
    for (uint32_t i = 0; i < own; i++)
        for (uint32_t j = 0; j <= i; j++)
            out_A[(size_t)i * (i + 1) / 2 + j] = (i == j ? diag[i] : C[(size_t)i * L + j]) & m;
    if (with_y)
        for (uint32_t i = 0; i < own; i++) out_b[i] = C[(size_t)own * L + i] & m;
    return LGC_OK;
}

// Per-thread device scratch for the per-pair calls below (a data provider runs one worker thread
// per peer; hipMalloc/hipFree per pair would dominate at small n).  Grown on demand, released
// when the thread exits.
struct P1Scratch {
    void *ptr[4];
    size_t cap[4];
    int device;
    P1Scratch() : device(-1) { for (int i = 0; i < 4; i++) { ptr[i] = 0; cap[i] = 0; } }
    ~P1Scratch() { release(); }
    void release() {
        for (int i = 0; i < 4; i++) { if (ptr[i]) (void)hipFree(ptr[i]); ptr[i] = 0; cap[i] = 0; }
    }
    hipError_t get(int dev, int slot, size_t bytes, void **out) {
        if (dev != device) { release(); device = dev; }
        if (cap[slot] < bytes) {
            if (ptr[slot]) (void)hipFree(ptr[slot]);
            ptr[slot] = 0; cap[slot] = 0;
            size_t want = bytes + bytes / 4 + 256;
            hipError_t e = hipMalloc(&ptr[slot], want);
            if (e != hipSuccess) return e;
            cap[slot] = want;
        }
        *out = ptr[slot];
        return hipSuccess;
    }
};
static thread_local P1Scratch t_scratch;

// Stream policy of the per-pair / per-batch calls.  A data provider runs a worker thread per peer, and five
// provider processes share one GPU in the single-box runs.  A stream (= a hardware queue) per worker thread
// oversubscribes the queues, which the scheduler then time-slices at millisecond granularity: config 4 takes
// 65.7 s that way against 21.5 s with all calls of a process taking turns on ONE stream -- measured again in
// round 2 with one call per batch of 16 pairs (per-thread streams: not kept).
#include <mutex>
static std::mutex g_p1_mutex;
static bool p1_shared_stream() { return true; }
struct P1Serial {
    bool locked;
    P1Serial() : locked(p1_shared_stream()) { if (locked) g_p1_mutex.lock(); }
    ~P1Serial() { if (locked) g_p1_mutex.unlock(); }
};
struct P1ThreadStream {
    hipStream_t st; int device;
    P1ThreadStream() : st(0), device(-1) {}
    ~P1ThreadStream() { if (st) (void)hipStreamDestroy(st); }
};
static thread_local P1ThreadStream t_stream;
static hipStream_t p1_stream() {
    if (p1_shared_stream()) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!t_stream.st || t_stream.device != dev) {
        if (t_stream.st) (void)hipStreamDestroy(t_stream.st);
        t_stream.st = 0; t_stream.device = dev;
        if (hipStreamCreateWithFlags(&t_stream.st, hipStreamNonBlocking) != hipSuccess) t_stream.st = 0;
    }
    return t_stream.st;
}

// out[q][k] = column cols[q] (d means y) +/- V[q][k]: the vectors a DP sends in inner_product_ti
// (b + x at phase1.c:201-207, a - y at 186-191).  Safe to call from several threads on one handle.
extern "C" int lgc_p1_mask(lgc_p1 *h, const uint32_t *cols, size_t npairs, const uint64_t *V, int sign, uint64_t *out) {
    if (!h || !cols || !V || !out) return lgc_fail(LGC_EINVAL, "null argument");
    if (npairs == 0) return LGC_OK;
    for (size_t q = 0; q < npairs; q++) if (cols[q] > h->d) return lgc_fail(LGC_EINVAL, "column out of range");
    P1CHK(hipSetDevice(h->device));
    P1Serial serial_; hipStream_t st = p1_stream();
    uint32_t *dcols = 0; uint64_t *dV = 0, *dout = 0;
    size_t bytes = npairs * h->n * sizeof(uint64_t);
    if (h->dev_io) {     // V and out are device memory (same-node rings): no copies
        P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
        P1CHK(hipMemcpyAsync(dcols, cols, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        unsigned gx = (unsigned)((h->n + 255) / 256); if (gx > 64) gx = 64;
        for (size_t q0 = 0; q0 < npairs; q0 += 65535) {
            unsigned gy = (unsigned)(npairs - q0 < 65535 ? npairs - q0 : 65535);
            hipLaunchKernelGGL(p1_mask_kernel, dim3(gx, gy), dim3(256), 0, st, h->X, h->n, h->d + 1, dcols + q0, V + q0 * h->n, sign, out + q0 * h->n, maskw(h->w));
            P1CHK(hipGetLastError());
        }
        P1CHK(hipGetLastError());
        P1CHK(hipStreamSynchronize(st));
        return LGC_OK;
    }
    if (npairs == 1) {   // per-pair protocol step: three operations
        P1CHK(t_scratch.get(h->device, 1, bytes, (void **)&dV));
        P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&dout));
        P1CHK(hipMemcpyAsync(dV, V, bytes, hipMemcpyHostToDevice, st));
        unsigned g1 = (unsigned)((h->n + 1023) / 1024); if (g1 > 64) g1 = 64;
        hipLaunchKernelGGL(p1_mask1_kernel, dim3(g1), dim3(1024), 0, st, h->X, h->n, h->d + 1, cols[0], dV, sign, dout, maskw(h->w));
        P1CHK(hipGetLastError());
        P1CHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
        P1CHK(hipStreamSynchronize(st));
        if (h->w == 32) for (size_t i = 0; i < h->n; i++) out[i] &= 0xffffffffull;
        return LGC_OK;
    }
    P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
    P1CHK(t_scratch.get(h->device, 1, bytes, (void **)&dV));
    P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&dout));
    P1CHK(hipMemcpyAsync(dcols, cols, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    P1CHK(hipMemcpyAsync(dV, V, bytes, hipMemcpyHostToDevice, st));
    unsigned gx = (unsigned)((h->n + 255) / 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(p1_mask_kernel, dim3(gx, (unsigned)npairs), dim3(256), 0, st, h->X, h->n, h->d + 1, dcols, dV, sign, dout, maskw(h->w));
    P1CHK(hipGetLastError());
    P1CHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
    P1CHK(hipStreamSynchronize(st));
    const uint64_t m = maskw(h->w);
    if (h->w == 32) for (size_t i = 0; i < npairs * h->n; i++) out[i] &= m;
    return LGC_OK;
}

// out[q] = <A[q], B[q]> (colsB == NULL) or <A[q], column colsB[q]>, minus sub[q]  (mod 2^w):
// the share arithmetic of inner_product_ti (phase1.c:194-196, 220-222).  Thread-safe like lgc_p1_mask.
extern "C" int lgc_p1_dot(lgc_p1 *h, const uint64_t *A, const uint64_t *B, const uint32_t *colsB, size_t npairs,
                          const uint64_t *sub, uint64_t *out) {
    if (!h || !A || !out || (!B && !colsB)) return lgc_fail(LGC_EINVAL, "null argument");
    if (npairs == 0) return LGC_OK;
    if (colsB) for (size_t q = 0; q < npairs; q++) if (colsB[q] > h->d) return lgc_fail(LGC_EINVAL, "column out of range");
    P1CHK(hipSetDevice(h->device));
    P1Serial serial_; hipStream_t st = p1_stream();
    size_t bytes = npairs * h->n * sizeof(uint64_t);
    uint64_t *dA = 0, *dB = 0, *dout = 0; uint32_t *dcols = 0;
    if (h->dev_io) {     // A (and B) are device memory; the sums still return to the host
        if (npairs > 65535) return lgc_fail(LGC_EINVAL, "too many pairs in one call");
        if (colsB) {
            P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
            P1CHK(hipMemcpyAsync(dcols, colsB, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
        P1CHK(t_scratch.get(h->device, 3, npairs * sizeof(uint64_t), (void **)&dout));
        P1CHK(hipMemsetAsync(dout, 0, npairs * sizeof(uint64_t), st));
        unsigned gx = (unsigned)((h->n + 255) / 256); if (gx > 64) gx = 64;
        hipLaunchKernelGGL(p1_dot_kernel, dim3(gx, (unsigned)npairs), dim3(256), 0, st, A, B, h->X, h->d + 1, dcols, h->n, dout);
        P1CHK(hipGetLastError());
        P1CHK(hipMemcpyAsync(out, dout, npairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        P1CHK(hipStreamSynchronize(st));
        const uint64_t mm = maskw(h->w);
        for (size_t q = 0; q < npairs; q++) out[q] = (out[q] - (sub ? sub[q] : 0)) & mm;
        return LGC_OK;
    }
    P1CHK(t_scratch.get(h->device, 1, bytes, (void **)&dA));
    P1CHK(hipMemcpyAsync(dA, A, bytes, hipMemcpyHostToDevice, st));
    if (npairs == 1) {   // per-pair protocol step: one workgroup writes the sum, no memset
        if (!colsB) {
            P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&dB));
            P1CHK(hipMemcpyAsync(dB, B, bytes, hipMemcpyHostToDevice, st));
        }
        P1CHK(t_scratch.get(h->device, 3, sizeof(uint64_t), (void **)&dout));
        P1CHK(hipMemsetAsync(dout, 0, sizeof(uint64_t), st));
        unsigned g1 = (unsigned)((h->n + 1023) / 1024); if (g1 > 64) g1 = 64;
        hipLaunchKernelGGL(p1_dot1_kernel, dim3(g1), dim3(1024), 0, st, dA, dB, h->X, h->d + 1, colsB ? colsB[0] : 0u, h->n, dout);
        P1CHK(hipGetLastError());
        P1CHK(hipMemcpyAsync(out, dout, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        P1CHK(hipStreamSynchronize(st));
        out[0] = (out[0] - (sub ? sub[0] : 0)) & maskw(h->w);
        return LGC_OK;
    }
    if (colsB) {
        P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
        P1CHK(hipMemcpyAsync(dcols, colsB, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    } else {
        P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&dB));
        P1CHK(hipMemcpyAsync(dB, B, bytes, hipMemcpyHostToDevice, st));
    }
    P1CHK(t_scratch.get(h->device, 3, npairs * sizeof(uint64_t), (void **)&dout));
    P1CHK(hipMemsetAsync(dout, 0, npairs * sizeof(uint64_t), st));
    unsigned gx = (unsigned)((h->n + 255) / 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(p1_dot_kernel, dim3(gx, (unsigned)npairs), dim3(256), 0, st, dA, dB, h->X, h->d + 1, dcols, h->n, dout);
    P1CHK(hipGetLastError());
    P1CHK(hipMemcpyAsync(out, dout, npairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    P1CHK(hipStreamSynchronize(st));
    const uint64_t m = maskw(h->w);
    for (size_t q = 0; q < npairs; q++) out[q] = (out[q] - (sub ? sub[q] : 0)) & m;
    return LGC_OK;
}

// Party a of one inner_product_ti (phase1.c:171-197) in a single pass: out_mask = a - y (sent to
// party b) and *share = <in, y> - sub, where `in` = b + x is party b's message, (y, sub) the TI's.
extern "C" int lgc_p1_ti_a(lgc_p1 *h, uint32_t col, const uint64_t *y, const uint64_t *in, uint64_t sub,
                           uint64_t *out_mask, uint64_t *share) {
    if (!h || !y || !in || !out_mask || !share) return lgc_fail(LGC_EINVAL, "null argument");
    if (col > h->d) return lgc_fail(LGC_EINVAL, "column out of range");
    P1CHK(hipSetDevice(h->device));
    P1Serial serial_; hipStream_t st = p1_stream();
    const size_t n = h->n, bytes = n * sizeof(uint64_t);
    uint64_t *dy = 0, *din = 0, *dout = 0;
    P1CHK(t_scratch.get(h->device, 1, bytes, (void **)&dy));
    P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&din));
    P1CHK(t_scratch.get(h->device, 3, bytes + sizeof(uint64_t), (void **)&dout));
    P1CHK(hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, st));
    P1CHK(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, st));
    P1CHK(hipMemsetAsync(dout + n, 0, sizeof(uint64_t), st));
    unsigned g1 = (unsigned)((n + 1023) / 1024); if (g1 > 64) g1 = 64;
    hipLaunchKernelGGL(p1_ti_a_kernel, dim3(g1), dim3(1024), 0, st, h->X, n, h->d + 1, col, dy, din, dout, maskw(h->w));
    P1CHK(hipGetLastError());
    P1CHK(hipMemcpyAsync(out_mask, dout, bytes, hipMemcpyDeviceToHost, st));
    uint64_t acc = 0;
    P1CHK(hipMemcpyAsync(&acc, dout + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    P1CHK(hipStreamSynchronize(st));
    const uint64_t m = maskw(h->w);
    if (h->w == 32) for (size_t i = 0; i < n; i++) out_mask[i] &= m;
    *share = (acc - sub) & m;
    return LGC_OK;
}

// Party a for a run of pairs with the same peer in one device call: y, in, out_mask are npairs x n words
// (page-locked host buffers move at the PCIe rate), cols / sub / shares have npairs entries.
extern "C" int lgc_p1_ti_a_batch(lgc_p1 *h, const uint32_t *cols, size_t npairs, const uint64_t *y, const uint64_t *in,
                                 const uint64_t *sub, uint64_t *out_mask, uint64_t *shares) {
    if (!h || !cols || !y || !in || !sub || !out_mask || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    if (npairs == 0) return LGC_OK;
    for (size_t q = 0; q < npairs; q++) if (cols[q] > h->d) return lgc_fail(LGC_EINVAL, "column out of range");
    P1CHK(hipSetDevice(h->device));
    P1Serial serial_; hipStream_t st = p1_stream();
    const size_t n = h->n, bytes = npairs * n * sizeof(uint64_t);
    uint64_t *dy = 0, *din = 0, *dout = 0; uint32_t *dcols = 0;
    if (h->dev_io) {     // y, in and out_mask are device memory
        uint64_t *dacc = 0;
        P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
        P1CHK(t_scratch.get(h->device, 3, npairs * sizeof(uint64_t), (void **)&dacc));
        P1CHK(hipMemcpyAsync(dcols, cols, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        P1CHK(hipMemsetAsync(dacc, 0, npairs * sizeof(uint64_t), st));
        unsigned gx = (unsigned)((n + 255) / 256); if (gx > 64) gx = 64;
        unsigned gy = npairs > 65535 ? 65535u : (unsigned)npairs;
        hipLaunchKernelGGL(p1_ti_a_batch_kernel, dim3(gx, gy), dim3(256), 0, st, h->X, n, h->d + 1, dcols, npairs, y, in, out_mask, dacc, maskw(h->w));
        P1CHK(hipGetLastError());
        P1CHK(hipMemcpyAsync(shares, dacc, npairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        P1CHK(hipStreamSynchronize(st));
        const uint64_t mm = maskw(h->w);
        for (size_t q = 0; q < npairs; q++) shares[q] = (shares[q] - sub[q]) & mm;
        return LGC_OK;
    }
    P1CHK(t_scratch.get(h->device, 0, npairs * sizeof(uint32_t), (void **)&dcols));
    P1CHK(t_scratch.get(h->device, 1, bytes, (void **)&dy));
    P1CHK(t_scratch.get(h->device, 2, bytes, (void **)&din));
    P1CHK(t_scratch.get(h->device, 3, bytes + npairs * sizeof(uint64_t), (void **)&dout));
    P1CHK(hipMemcpyAsync(dcols, cols, npairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    P1CHK(hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, st));
    P1CHK(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, st));
    P1CHK(hipMemsetAsync(dout + npairs * n, 0, npairs * sizeof(uint64_t), st));
    unsigned gx = (unsigned)((n + 255) / 256); if (gx > 64) gx = 64;
    unsigned gy = npairs > 65535 ? 65535u : (unsigned)npairs;
    hipLaunchKernelGGL(p1_ti_a_batch_kernel, dim3(gx, gy), dim3(256), 0, st, h->X, n, h->d + 1, dcols, npairs, dy, din, dout, dout + npairs * n, maskw(h->w));
    P1CHK(hipGetLastError());
    P1CHK(hipMemcpyAsync(out_mask, dout, bytes, hipMemcpyDeviceToHost, st));
    P1CHK(hipMemcpyAsync(shares, dout + npairs * n, npairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    P1CHK(hipStreamSynchronize(st));
    const uint64_t m = maskw(h->w);
    if (h->w == 32) for (size_t i = 0; i < npairs * n; i++) out_mask[i] &= m;
    for (size_t q = 0; q < npairs; q++) shares[q] = (shares[q] - sub[q]) & m;
    return LGC_OK;
}

extern "C" int lgc_p1_set_device_io(lgc_p1 *h, int on) {
    if (!h) return lgc_fail(LGC_EINVAL, "null handle");
    h->dev_io = on != 0;
    return LGC_OK;
}

// the TI's keystream split straight into per-pair destinations (device rings of the data providers): word t of
// pair q sits at stream word q (2n + 1) + t; x -> xdst[q], y -> ydst[q], r -> r[q].  Aligned word loads
// (the stream offset of a batch is a multiple of the word size).
template <typename WT>
__global__ void ti_unpack_scatter_kernel(const WT *ks, size_t npairs, size_t n, uint64_t *const *xdst, uint64_t *const *ydst, uint64_t *r) {
    const size_t per = 2 * n + 1;
    for (size_t q = blockIdx.y; q < npairs; q += gridDim.y) {
        const WT *src = ks + q * per;
        uint64_t *dx = xdst[q], *dy = ydst[q];
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < per; t += (size_t)gridDim.x * blockDim.x) {
            const uint64_t v = (uint64_t)src[t];
            if (t < n) dx[t] = v;
            else if (t < 2 * n) dy[t - n] = v;
            else r[q] = v;
        }
    }
}
// out[q] = <xdst[q], ydst[q]>  (mod 2^64); out zeroed by the host
__global__ void __launch_bounds__(256)
p1_dot_ptr_kernel(uint64_t *const *xdst, uint64_t *const *ydst, size_t npairs, size_t n, uint64_t *out) {
    for (size_t q = blockIdx.y; q < npairs; q += gridDim.y) {
        const uint64_t *a = xdst[q], *b = ydst[q];
        uint64_t acc = 0;
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) acc += a[k] * b[k];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&out[q], (unsigned long long)acc);
    }
}

// Trusted initializer (phase1.c:241-287): for pairs [first_pair, first_pair + npairs) of the
// (i, j) enumeration, words x[n], y[n], r drawn in this order from one AES-128-CTR stream keyed
// by `seed`; xy_minus_r[q] = <x,y> - r.  Stream position of pair q is q * (2n + 1) words.
static int ti_generate(int device, const uint8_t seed[16], uint64_t first_pair, size_t npairs, size_t n, int width,
                       uint64_t *x, uint64_t *y, void *const *x_dst, void *const *y_dst, uint64_t *r, uint64_t *xy_minus_r) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!seed || !r || !xy_minus_r || (!x && !x_dst) || (!y && !y_dst)) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    if (npairs == 0) return LGC_OK;
    int rc = lgc_need_device(device);
    if (rc) return rc;
    AesTables t;
    aes_build_tables(t, seed);
    uint32_t *drk = 0;
    P1CHK(hipMalloc(&drk, sizeof(t.rk))); dev_guard.add(drk);
    P1CHK(hipMemcpy(drk, t.rk, sizeof(t.rk), hipMemcpyHostToDevice));
    const size_t wb = width / 8;
    const uint64_t words_per_pair = 2 * n + 1;
    const uint64_t byte0 = first_pair * words_per_pair * wb, byte1 = (first_pair + npairs) * words_per_pair * wb;
    const uint64_t blk0 = byte0 / 16, blk1 = (byte1 + 15) / 16;
    uint4 *dks = 0;
    P1CHK(hipMalloc(&dks, (blk1 - blk0) * 16)); dev_guard.add(dks);
    hipLaunchKernelGGL(ti_prg_kernel, dim3(512), dim3(1024), 0, 0, drk, blk0, blk1 - blk0, dks);
    P1CHK(hipGetLastError());
    const uint64_t m = maskw(width);
    uint64_t *dA = 0, *dB = 0, *dr = 0, *dout = 0;
    size_t bytes = npairs * n * 8;
    P1CHK(hipMalloc(&dA, bytes)); dev_guard.add(dA); P1CHK(hipMalloc(&dB, bytes)); dev_guard.add(dB); P1CHK(hipMalloc(&dr, npairs * 8)); dev_guard.add(dr); P1CHK(hipMalloc(&dout, npairs * 8)); dev_guard.add(dout);
    hipLaunchKernelGGL(ti_unpack_kernel, dim3(1024), dim3(256), 0, 0, (const uint8_t *)dks, (size_t)(byte0 - blk0 * 16), npairs, n,
                       (int)wb, dA, dB, dr);
    P1CHK(hipGetLastError());
    P1CHK(hipMemset(dout, 0, npairs * 8));
    unsigned gx = (unsigned)((n + 255) / 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(p1_dot_kernel, dim3(gx, (unsigned)npairs), dim3(256), 0, 0, dA, dB, (const int64_t *)0, (size_t)0,
                       (const uint32_t *)0, n, dout);
    P1CHK(hipGetLastError());
    if (x_dst) {
        return lgc_fail(LGC_EINVAL, "internal: scatter requests go through ti_generate_scatter");
    } else {
        P1CHK(hipMemcpy(x, dA, bytes, hipMemcpyDeviceToHost));
        P1CHK(hipMemcpy(y, dB, bytes, hipMemcpyDeviceToHost));
    }
    P1CHK(hipMemcpy(r, dr, npairs * 8, hipMemcpyDeviceToHost));
    P1CHK(hipMemcpy(xy_minus_r, dout, npairs * 8, hipMemcpyDeviceToHost));

    for (size_t q = 0; q < npairs; q++) xy_minus_r[q] = (xy_minus_r[q] - r[q]) & m;
    return LGC_OK;
}
extern "C" int lgc_ti_generate(int device, const uint8_t seed[16], uint64_t first_pair, size_t npairs, size_t n, int width,
                               uint64_t *x, uint64_t *y, uint64_t *r, uint64_t *xy_minus_r) {
    if (!x || !y) return lgc_fail(LGC_EINVAL, "null argument");
    return ti_generate(device, seed, first_pair, npairs, n, width, x, y, 0, 0, r, xy_minus_r);
}
// the same, x[q] / y[q] written to the device addresses x_dst[q] / y_dst[q] (host arrays of device pointers)
extern "C" int lgc_ti_generate_scatter(int device, const uint8_t seed[16], uint64_t first_pair, size_t npairs, size_t n, int width,
                                       void *const *x_dst, void *const *y_dst, uint64_t *r, uint64_t *xy_minus_r) {
    if (!x_dst || !y_dst || !seed || !r || !xy_minus_r) return lgc_fail(LGC_EINVAL, "null argument");
    if (width != 32 && width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    if (npairs == 0) return LGC_OK;
    int rc = lgc_need_device(device);
    if (rc) return rc;
    // keystream -> destinations -> <x, y>: three kernels on grow-only per-thread scratch, no allocation per batch
    AesTables t;
    aes_build_tables(t, seed);
    const size_t wb = width / 8;
    const uint64_t words_per_pair = 2 * n + 1;
    const uint64_t byte0 = first_pair * words_per_pair * wb, byte1 = (first_pair + npairs) * words_per_pair * wb;
    const uint64_t blk0 = byte0 / 16, blk1 = (byte1 + 15) / 16;
    char *small = 0; uint4 *dks = 0;
    const size_t ptr_bytes = 2 * npairs * sizeof(void *), small_bytes = 256 + ptr_bytes + 2 * npairs * 8;
    P1CHK(t_scratch.get(device, 0, small_bytes, (void **)&small));
    P1CHK(t_scratch.get(device, 1, (blk1 - blk0) * 16, (void **)&dks));
    uint32_t *drk = reinterpret_cast<uint32_t *>(small);
    uint64_t **dptr = reinterpret_cast<uint64_t **>(small + 256);
    uint64_t *dr = reinterpret_cast<uint64_t *>(small + 256 + ptr_bytes), *dout = dr + npairs;
    P1CHK(hipMemcpyAsync(drk, t.rk, sizeof(t.rk), hipMemcpyHostToDevice, 0));
    P1CHK(hipMemcpyAsync(dptr, x_dst, npairs * sizeof(void *), hipMemcpyHostToDevice, 0));
    P1CHK(hipMemcpyAsync(dptr + npairs, y_dst, npairs * sizeof(void *), hipMemcpyHostToDevice, 0));
    P1CHK(hipMemsetAsync(dout, 0, npairs * 8, 0));
    hipLaunchKernelGGL(ti_prg_kernel, dim3(512), dim3(1024), 0, 0, drk, blk0, blk1 - blk0, dks);
    P1CHK(hipGetLastError());
    const char *ks0 = reinterpret_cast<const char *>(dks) + (byte0 - blk0 * 16);
    unsigned gx = (unsigned)((2 * n + 1 + 255) / 256); if (gx > 128) gx = 128;
    unsigned gy = npairs > 65535 ? 65535u : (unsigned)npairs;
    if (wb == 8) hipLaunchKernelGGL((ti_unpack_scatter_kernel<uint64_t>), dim3(gx, gy), dim3(256), 0, 0, (const uint64_t *)ks0, npairs, n, dptr, dptr + npairs, dr);
    else hipLaunchKernelGGL((ti_unpack_scatter_kernel<uint32_t>), dim3(gx, gy), dim3(256), 0, 0, (const uint32_t *)ks0, npairs, n, dptr, dptr + npairs, dr);
    unsigned gd = (unsigned)((n + 255) / 256); if (gd > 64) gd = 64;
    hipLaunchKernelGGL(p1_dot_ptr_kernel, dim3(gd, gy), dim3(256), 0, 0, dptr, dptr + npairs, npairs, n, dout);
    P1CHK(hipGetLastError());
    P1CHK(hipMemcpy(r, dr, npairs * 8, hipMemcpyDeviceToHost));
    P1CHK(hipMemcpy(xy_minus_r, dout, npairs * 8, hipMemcpyDeviceToHost));
    const uint64_t m = maskw(width);
    for (size_t q = 0; q < npairs; q++) xy_minus_r[q] = (xy_minus_r[q] - r[q]) & m;
    return LGC_OK;
}
