// gc_device.h -- CDNA4 (gfx950) wave backends for the garbled word machine.
//
// One wavefront executes one record; lane l holds the 128-bit label of bit l.
//   * labels move HBM <-> VGPR as one global_load/store_dwordx4 per lane
//     (1 KiB contiguous per word per wave)
//   * the AES T-table is staged once per workgroup in LDS, 32x replicated so
//     that every ds_read_b32 is bank-conflict-free
//   * garbled tables are written/read as two 1 KiB coalesced rows per step
//   * lane moves are ds_bpermute / v_readlane; public lane masks are SGPRs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gc_aes.h"
#include "gc_exec.h"

namespace gc {

__constant__ uint32_t c_rk[44];
__constant__ uint32_t c_te0[256];

static constexpr int kLdsTabWords = 256 * 32;   // 32 KiB

struct LdsTab {
    const uint32_t *base;   // LDS, already offset by (lane & 31)
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i << 5]; }
};

__device__ __forceinline__ void lds_tab_fill(uint32_t *lds) {
    for (int i = threadIdx.x; i < kLdsTabWords; i += blockDim.x) lds[i] = c_te0[i >> 5];
    __syncthreads();
}

__device__ __forceinline__ Lbl ld_lbl(const Lbl *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    Lbl r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ void st_lbl(Lbl *p, Lbl v) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(v.x, v.y, v.z, v.w);
}

template <bool GARBLER, bool INLINE_AND>
struct GpuBackend {
    typedef Lbl W;
    Lbl R;               // garbler's global offset (lsb = 1); unused by the evaluator
    Lbl *words;          // word file
    Lbl *tab;            // this launch's table buffer
    uint64_t *decode;    // reveal slots
    uint64_t step;       // global gate-step counter (wave-uniform)
    uint64_t launch_step0;
    int lane;
    LdsTab lt;

    __device__ __forceinline__ W zero() const { return lzero(); }
    __device__ __forceinline__ uint32_t bit(uint64_t m) const { return (uint32_t)(m >> lane) & 1u; }
    __device__ __forceinline__ W konst(uint64_t bits) const { return GARBLER ? lmask(R, bit(bits)) : lzero(); }
    __device__ __forceinline__ W XOR(W a, W b) const { return lxor(a, b); }
    __device__ __forceinline__ W NOTm(W a, uint64_t m) const { return GARBLER ? lxor(a, lmask(R, bit(m))) : a; }
    __device__ __forceinline__ W sel(uint64_t m, W a, W b) const {
        bool t = bit(m);
        W r = {t ? a.x : b.x, t ? a.y : b.y, t ? a.z : b.z, t ? a.w : b.w};
        return r;
    }
    __device__ __forceinline__ W bcast(W a, int src) const {
        W r = {(uint32_t)__builtin_amdgcn_readlane((int)a.x, src), (uint32_t)__builtin_amdgcn_readlane((int)a.y, src),
               (uint32_t)__builtin_amdgcn_readlane((int)a.z, src), (uint32_t)__builtin_amdgcn_readlane((int)a.w, src)};
        return r;
    }
    __device__ __forceinline__ W pull(W a, int from, bool ok) const {
        int addr = (from & 63) << 2;
        W r = {(uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.x), (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.y),
               (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.z), (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.w)};
        uint32_t k = ok ? 0xffffffffu : 0u;
        r.x &= k; r.y &= k; r.z &= k; r.w &= k;
        return r;
    }
    // lane l <- lane l-k (zero fill)
    __device__ __forceinline__ W shl(W a, int k) const {
        if (k >= 64) return lzero();
        return pull(a, lane - k, lane >= k);
    }
    // lane l <- lane l+k (zero fill)
    __device__ __forceinline__ W shr(W a, int k) const {
        if (k >= 64) return lzero();
        return pull(a, lane + k, lane + k < 64);
    }
    __device__ __forceinline__ W AND(W a, W b, uint64_t act) {
        const bool on = bit(act);
        const uint64_t gid = step * 64 + (uint64_t)lane;
        Lbl *slot = tab + (step - launch_step0) * 128 + lane;
        step++;
        if (INLINE_AND) return and_impl(lt, R, a, b, gid, slot, on);
        return and_outlined(lt, R, a, b, gid, slot, on);
    }
    static __device__ __forceinline__ W and_impl(LdsTab lt, Lbl R, W a, W b, uint64_t gid, Lbl *slot, bool on) {
        W c = lzero();
        if (on) {
            if (GARBLER) {
                Lbl TG, TE;
                c = garble_and(lt, c_rk, R, a, b, gid, TG, TE);
                st_lbl(slot, TG);
                st_lbl(slot + 64, TE);
            } else {
                Lbl TG = ld_lbl(slot), TE = ld_lbl(slot + 64);
                c = eval_and(lt, c_rk, a, b, gid, TG, TE);
            }
        }
        return c;
    }
    // the generic (non-MAC) kernel keeps one copy of the gate body: code size, compile time
    static __device__ __noinline__ W and_outlined(LdsTab lt, Lbl R, W a, W b, uint64_t gid, Lbl *slot, bool on) {
        return and_impl(lt, R, a, b, gid, slot, on);
    }
    __device__ __forceinline__ W load(uint32_t id) const { return ld_lbl(words + (size_t)id * 64 + lane); }
    __device__ __forceinline__ void store(uint32_t id, W v) { st_lbl(words + (size_t)id * 64 + lane, v); }
    __device__ __forceinline__ void reveal(uint32_t slot, W v) {
        uint64_t m = __ballot(v.x & 1u);
        if (lane == 0) decode[slot] = m;
    }
};

// one wavefront per record; 4 waves per workgroup share the LDS table
template <bool GARBLER, bool MAC_ONLY>
__global__ void __launch_bounds__(256)
gc_exec_kernel(const Rec *recs, uint32_t nrec, Lbl *words, Lbl *tab, uint64_t *decode,
               uint64_t launch_step0, Lbl R, int w, int p) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    lds_tab_fill(lds_te0);
    const int lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= nrec) return;
    GpuBackend<GARBLER, MAC_ONLY> be;
    be.R = R;
    be.words = words;
    be.tab = tab;
    be.decode = decode;
    be.launch_step0 = launch_step0;
    be.lane = lane;
    be.lt.base = lds_te0 + (lane & 31);
    Rec r = recs[wid];
    // make the record wave-uniform for the compiler (SGPRs)
    r.op = __builtin_amdgcn_readfirstlane(r.op);
    r.cnt = __builtin_amdgcn_readfirstlane(r.cnt);
    r.dst = __builtin_amdgcn_readfirstlane(r.dst);
    r.a = __builtin_amdgcn_readfirstlane(r.a);
    r.b = __builtin_amdgcn_readfirstlane(r.b);
    r.c = __builtin_amdgcn_readfirstlane(r.c);
    r.sa = __builtin_amdgcn_readfirstlane(r.sa);
    r.sb = __builtin_amdgcn_readfirstlane(r.sb);
    uint32_t s_lo = __builtin_amdgcn_readfirstlane((uint32_t)r.step0);
    uint32_t s_hi = __builtin_amdgcn_readfirstlane((uint32_t)(r.step0 >> 32));
    be.step = ((uint64_t)s_hi << 32) | s_lo;
    if (MAC_ONLY) {
        if (r.op == OP_MAC) {
            typedef Circ<GpuBackend<GARBLER, MAC_ONLY>> C;
            Lbl S = lzero(), Cc = lzero();
            for (uint32_t k = 0; k < r.cnt; k++)
                C::mac(be, S, Cc, be.load(r.a + (int32_t)k * r.sa), be.load(r.b + (int32_t)k * r.sb), w, p);
            be.store(r.dst, S);
            be.store(r.dst + 1, Cc);
        }
    } else {
        exec_record(be, r, w, p);
    }
}

}  // namespace gc
