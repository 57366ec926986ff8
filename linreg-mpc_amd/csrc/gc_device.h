// gc_device.h -- CDNA4 (gfx950) wave backends for the garbled word machine.
//
// One wavefront executes one record; lane l holds the 128-bit label of bit l.
//   * labels move HBM <-> VGPR as one global_load/store_dwordx4 per lane
//     (1 KiB contiguous per word per wave)
//   * the AES T-table is staged once per workgroup in LDS, 64x replicated so
//     that every ds_read_b32 is bank-conflict-free and addressed by one v_perm
//   * garbled tables are written/read as two 1 KiB coalesced rows per step
//   * lane moves are ds_bpermute / v_readlane; public lane masks are SGPRs
//   * narrow launches run one record per 4-wave workgroup, the waves splitting the AES work
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gc_aes.h"
#include "gc_exec.h"

namespace gc {

// Device constants of the fixed-key AES: compile-time data (gc_aes.h aes_make_dev_const).  `static`: every translation
// unit of the library (the garbler kernels, the evaluator kernels, the engine, phase 1, OT -- compiled in parallel,
// csrc/Makefile) has its own copy in its own code object.  Rounds 1-3 filled them with hipMemcpyToSymbol when a process
// touched the device for the first time, which loaded ALL code objects of the library into every process -- a data
// provider that runs four phase-1 kernels paid for the garbler's and the evaluator's MAC kernels.
static __constant__ DevAesConst c_aes = aes_make_dev_const();

static constexpr int kLdsTabWords = 256 * 64;   // 64 KiB: entry x occupies the 256-byte row x

struct LdsTab {
    static const bool kTwoTables = false;
    static const bool kFourTables = false;
    const char *base;    // LDS byte address of the table
    uint32_t lane4;      // (lane << 2): fits one byte, merged into the address by v_perm_b32
    // Te0[byte k of word]: address = (byte << 8) | (lane << 2)
    __device__ __forceinline__ uint32_t lk(uint32_t word, int k) const {
        uint32_t off = __builtin_amdgcn_perm(word, lane4, 0x0c0c0400u + ((uint32_t)k << 8));
        return *reinterpret_cast<const uint32_t *>(base + off);
    }
    __device__ __forceinline__ uint32_t lk2(uint32_t word, int k) const { return rotl32(lk(word, k), 16); }
    __device__ __forceinline__ uint32_t lkt(int, uint32_t word, int k) const { return lk(word, k); }
};

// ---- staging the AES table image.  Row x of an image is 64 dwords that depend on Te0[x] and the position in the row only, so
// wave w of a workgroup writes rows w, w + W, w + 2W, ...: Te0[x] is wave-uniform -- a SCALAR load, four in flight --, a row is
// one ds_write_b32 of the whole wave, and nothing waits for vector memory.  (Rounds 1-4 had thread i write dwords i,
// i + blockDim, ...: one vector load of Te0 per dword, each waited for before its store, sixteen times in a row.  Measured in
// round 5: no difference in any launch profile -- those loads hit the L1 after the first workgroup -- so this is tidiness,
// not speed.)
template <class F>
__device__ __forceinline__ void lds_tab_rows(F row) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), waves = blockDim.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    // four rows per trip, their four scalar loads issued together (indices clamped, stores guarded: no trip-count arithmetic)
    for (uint32_t x = wave; x < 256u; x += 4u * waves) {
        const uint32_t x1 = x + waves, x2 = x1 + waves, x3 = x2 + waves;
        const uint32_t v0 = c_aes.te0[x], v1 = c_aes.te0[x1 < 256u ? x1 : 255u], v2 = c_aes.te0[x2 < 256u ? x2 : 255u],
                       v3 = c_aes.te0[x3 < 256u ? x3 : 255u];
        row(x, lane, v0);
        if (x1 < 256u) row(x1, lane, v1);
        if (x2 < 256u) row(x2, lane, v2);
        if (x3 < 256u) row(x3, lane, v3);
    }
    __syncthreads();
}
// Four tables Te_t = rotl(Te0, 8t) in the same 128 KiB: a ds_read_b32 is serviced in two groups of
// 32 lanes on 32 banks, so 32 replicas per entry are already conflict-free (lanes l and l + 32 share
// a replica but never a cycle).  Row x (256 B) of the first half holds Te0[x] x 32 | Te1[x] x 32,
// of the second half Te2[x] x 32 | Te3[x] x 32.  No rotates in the rounds.
struct LdsTab4 {
    static const bool kTwoTables = false;
    static const bool kFourTables = true;
    const char *base;
    uint32_t c[4];       // per table: ((lane & 31) << 2) | (t & 1) << 7 | (t >> 1) << 16
    __device__ __forceinline__ uint32_t lk(uint32_t word, int k) const { return lkt(0, word, k); }
    __device__ __forceinline__ uint32_t lk2(uint32_t word, int k) const { return lkt(2, word, k); }
    __device__ __forceinline__ uint32_t lkt(int t, uint32_t word, int k) const {
        uint32_t off = __builtin_amdgcn_perm(word, c[t], (t >= 2 ? 0x0c020400u : 0x0c0c0400u) + ((uint32_t)k << 8));
        return *reinterpret_cast<const uint32_t *>(base + off);
    }
};
__device__ __forceinline__ void lds_tab4_fill(uint32_t *lds) {
    lds_tab_rows([lds](uint32_t x, uint32_t lane, uint32_t v) {
        const bool odd = lane >= 32u;            // second half of a row: the table rotated by one more byte
        lds[x * 64 + lane] = odd ? ((v << 8) | (v >> 24)) : v;
        lds[kLdsTabWords + x * 64 + lane] = odd ? ((v << 24) | (v >> 8)) : ((v << 16) | (v >> 16));
    });
}
__device__ __forceinline__ LdsTab4 lds_tab4_make(const uint32_t *lds) {
    LdsTab4 t;
    t.base = reinterpret_cast<const char *>(lds);
    const uint32_t l5 = (threadIdx.x & 31u) << 2;
    t.c[0] = l5; t.c[1] = l5 | 0x80u; t.c[2] = l5 | 0x10000u; t.c[3] = l5 | 0x10080u;
    return t;
}

__device__ __forceinline__ void lds_tab_fill(uint32_t *lds) {
    lds_tab_rows([lds](uint32_t x, uint32_t lane, uint32_t v) { lds[x * 64 + lane] = v; });
}
__device__ __forceinline__ LdsTab lds_tab_make(const uint32_t *lds) {
    LdsTab t;
    t.base = reinterpret_cast<const char *>(lds);
    t.lane4 = (threadIdx.x & 63u) << 2;
    return t;
}

// Two tables in 64 KiB (32 replicas): row x = Te0[x] x 32 | Te2[x] x 32.  One rotate per column
// (as LdsTab2) at half the footprint, so two 4-wave workgroups still fit a CU.
struct LdsTab2h {
    static const bool kTwoTables = true;
    static const bool kFourTables = false;
    const char *base;
    uint32_t c0, c1;
    __device__ __forceinline__ uint32_t lk(uint32_t word, int k) const {
        uint32_t off = __builtin_amdgcn_perm(word, c0, 0x0c0c0400u + ((uint32_t)k << 8));
        return *reinterpret_cast<const uint32_t *>(base + off);
    }
    __device__ __forceinline__ uint32_t lk2(uint32_t word, int k) const {
        uint32_t off = __builtin_amdgcn_perm(word, c1, 0x0c0c0400u + ((uint32_t)k << 8));
        return *reinterpret_cast<const uint32_t *>(base + off);
    }
    __device__ __forceinline__ uint32_t lkt(int, uint32_t word, int k) const { return lk(word, k); }
};
__device__ __forceinline__ void lds_tab2h_fill(uint32_t *lds) {
    lds_tab_rows([lds](uint32_t x, uint32_t lane, uint32_t v) { lds[x * 64 + lane] = (lane >= 32u) ? ((v << 16) | (v >> 16)) : v; });
}
__device__ __forceinline__ LdsTab2h lds_tab2h_make(const uint32_t *lds) {
    LdsTab2h t;
    t.base = reinterpret_cast<const char *>(lds);
    t.c0 = (threadIdx.x & 31u) << 2;
    t.c1 = t.c0 | 0x80u;
    return t;
}

// table variant by number: 2 = LdsTab2h (64 KiB), 4 = LdsTab4 (128 KiB)
template <int TABV> struct TabSel;
template <> struct TabSel<2> {
    typedef LdsTab2h T;
    static constexpr int kWords = kLdsTabWords;
    static __device__ __forceinline__ void fill(uint32_t *lds) { lds_tab2h_fill(lds); }
    static __device__ __forceinline__ T make(const uint32_t *lds) { return lds_tab2h_make(lds); }
};
template <> struct TabSel<4> {
    typedef LdsTab4 T;
    static constexpr int kWords = 2 * kLdsTabWords;
    static __device__ __forceinline__ void fill(uint32_t *lds) { lds_tab4_fill(lds); }
    static __device__ __forceinline__ T make(const uint32_t *lds) { return lds_tab4_make(lds); }
};

__device__ __forceinline__ Lbl ld_lbl(const Lbl *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    Lbl r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ void st_lbl(Lbl *p, Lbl v) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(v.x, v.y, v.z, v.w);
}
// The same store with the address space spelled out.  Where one branch of a wave-uniform `if` stores a label to
// the table buffer and the other one to the LDS exchange area, the compiler otherwise sinks the two stores into ONE
// flat_store (seen in the 4-wave critical-path step): a flat store counts on lgkmcnt, so the wait in front of the
// workgroup barrier then sat out a global-memory round trip on every gate level (15 % of a divider's time).
typedef uint32_t gc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_lbl_global(Lbl *p, Lbl v) {
    gc_u32x4 d = {v.x, v.y, v.z, v.w};
    *(__attribute__((address_space(1))) gc_u32x4 *)p = d;
}
__device__ __forceinline__ Lbl ld_lbl_global(const Lbl *p) {
    gc_u32x4 v = *(const __attribute__((address_space(1))) gc_u32x4 *)p;
    Lbl r = {v.x, v.y, v.z, v.w};
    return r;
}
// garbled-table rows are written once and read once, a launch (tens of GiB) later: non-temporal, so that they do not push the
// word file and the scratch lines out of L2 on their way through
__device__ __forceinline__ void st_lbl_global_nt(Lbl *p, Lbl v) {
    gc_u32x4 d = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(d, (__attribute__((address_space(1))) gc_u32x4 *)p);
}
__device__ __forceinline__ Lbl ld_lbl_global_nt(const Lbl *p) {
    gc_u32x4 v = __builtin_nontemporal_load((const __attribute__((address_space(1))) gc_u32x4 *)p);
    Lbl r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ void st_lbl_lds(Lbl *p, Lbl v) {
    gc_u32x4 d = {v.x, v.y, v.z, v.w};
    *(__attribute__((address_space(3))) gc_u32x4 *)p = d;
}

// MODE_MAC : one wave does all AES of its gate step, gate body inlined (MAC kernels).
// MODE_SOLO: the same for generic records in WIDE launches (thousands of records: throughput matters, not latency).
// MODE_QUAD: the 4 waves of a workgroup run the same record redundantly and split the
//            4 (garbler) / 2 (evaluator) hashes of every gate step between them, exchanging
//            the results through LDS: the latency of a dependent chain of steps drops from
//            4 sequential AES to 1 AES + one barrier.  Used for the narrow, latency-bound
//            launches (dividers, adders, max trees).
enum { MODE_MAC = 0, MODE_SOLO = 1, MODE_QUAD = 2 };

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the
// vector-memory queue (vmcnt), so table stores / prefetched table loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");   // no LDS access of the next phase may be scheduled above the barrier
}

template <bool GARBLER, int MODE, class TAB = LdsTab>
struct GpuBackend {
    typedef Lbl W;
    // latency-bound kernels issue independent gate steps of the multiplier as dual steps (gc_circuits.h)
    static const bool kPairSteps = (MODE == MODE_QUAD);
    int wave;            // MODE_QUAD: wave index inside the workgroup (wave-uniform)
    Lbl *xch;            // MODE_QUAD: LDS exchange area, 2 buffers x 512 labels (16 KiB)
    int xsel;            // MODE_QUAD: buffer used by the next step (adjacent steps alternate)
    Lbl R;               // garbler's global offset (lsb = 1); unused by the evaluator
    Lbl *words;          // word file
    Lbl *tab;            // this launch's table buffer
    uint64_t *decode;    // reveal slots
    uint64_t step;       // global gate-step counter (wave-uniform)
    uint64_t launch_step0;
    int lane;
    TAB lt;

    __device__ __forceinline__ W zero() const { return lzero(); }
    // public lane masks are wave-uniform 64-bit scalars: used directly as the v_cndmask condition
    __device__ __forceinline__ bool bit(uint64_t m) const { return __builtin_amdgcn_inverse_ballot_w64(m); }
    __device__ __forceinline__ W rmask(uint64_t m) const {
        bool t = bit(m);
        W r = {t ? R.x : 0u, t ? R.y : 0u, t ? R.z : 0u, t ? R.w : 0u};
        return r;
    }
    __device__ __forceinline__ W konst(uint64_t bits) const { return GARBLER ? rmask(bits) : lzero(); }
    __device__ __forceinline__ W XOR(W a, W b) const { return lxor(a, b); }
    __device__ __forceinline__ W NOTm(W a, uint64_t m) const { return GARBLER ? lxor(a, rmask(m)) : a; }
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
    __device__ __forceinline__ W bcast2(W a, int r) const {
        W lo = bcast(a, r), hi = bcast(a, 32 + r);
        return sel(0xffffffffull, lo, hi);
    }
    __device__ __forceinline__ W pull(W a, int from, bool ok) const {
        int addr = (from & 63) << 2;
        W r = {(uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.x), (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.y),
               (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.z), (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)a.w)};
        uint32_t k = ok ? 0xffffffffu : 0u;
        r.x &= k; r.y &= k; r.z &= k; r.w &= k;
        return r;
    }
    // every lane <- the top lane of the lower half of its block of 2^(k+1) lanes (Circ::add: the fan-out of a prefix node)
    __device__ __forceinline__ W bblk(W a, int k) const {
        const int h = 1 << k;
        return pull(a, (lane & ~(2 * h - 1)) | (h - 1), true);
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
        if (MODE == MODE_SOLO || MODE == MODE_MAC) return and_impl(lt, R, a, b, gid, slot, on);
        xsel ^= 1;
        return and_quad(lt, R, a, b, gid, slot, on, wave, xch + xsel * 512, lane);
    }
    __device__ __forceinline__ void AND2(W a1, W b1, uint64_t act1, W a2, W b2, uint64_t act2, W &c1, W &c2) {
        if (MODE != MODE_QUAD) {
            c1 = AND(a1, b1, act1);
            c2 = AND(a2, b2, act2);
            return;
        }
        const bool on1 = bit(act1), on2 = bit(act2);
        const uint64_t gid = step * 64 + (uint64_t)lane;
        Lbl *slot = tab + (step - launch_step0) * 128 + lane;
        step += 2;
        xsel ^= 1;
        and2_quad(lt, R, a1, b1, a2, b2, gid, slot, on1, on2, wave, xch + xsel * 512, lane, c1, c2);
    }
    static __device__ __forceinline__ W and_impl(TAB lt, Lbl R, W a, W b, uint64_t gid, Lbl *slot, bool on) {
        W c = lzero();
        if (on) {
            if (GARBLER) {
                Lbl TG, TE;
                c = garble_and(lt, c_aes.rk, R, a, b, gid, TG, TE, c_aes.rk24);
                st_lbl_global_nt(slot, TG);
                st_lbl_global_nt(slot + 64, TE);
            } else {
                Lbl TG = ld_lbl_global_nt(slot), TE = ld_lbl_global_nt(slot + 64);
                c = eval_and(lt, c_aes.rk, a, b, gid, TG, TE, c_aes.rk24);
            }
        }
        return c;
    }
    // cooperative gate step.  Every wave of the workgroup calls this with identical operands; exactly
    // one barrier per step, the exchange area is double-buffered by step parity.  Inlined: the call
    // ABI spilled around every gate step (DIV 3.5 -> 2.4 ms).
    // Evaluator: the two ciphertext rows are fetched by ALL lanes before the hash (rows are 64 labels
    // wide whatever the activity mask), so no EXEC-masked load forces a vmcnt(0) ahead of the hash.
    static __device__ __forceinline__ W and_quad(TAB lt, Lbl R, W a, W b, uint64_t gid, Lbl *slot, bool on,
                                                 int wave, Lbl *xbuf, int lane) {
        const int nh = GARBLER ? 4 : 2;
        Lbl TGe = lzero(), TEe = lzero();
        if (!GARBLER) { TGe = ld_lbl(slot); TEe = ld_lbl(slot + 64); }   // in flight during the hash
        if (wave < nh) {
            Lbl h = lzero();
            if (on) {
                Lbl x = (wave < (nh >> 1)) ? a : b;
                if (GARBLER && (wave & 1)) x = lxor(x, R);
                uint64_t tw = 2 * gid + (uint64_t)(wave >= (nh >> 1));
                hash_n<1, TAB>(lt, c_aes.rk, &x, &tw, &h, c_aes.rk24);
            }
            xbuf[wave * 64 + lane] = h;
        }
        lds_barrier();
        W c = lzero();
        if (GARBLER) {
            if (on) {
                Lbl h0 = xbuf[lane], h1 = xbuf[64 + lane], h2 = xbuf[128 + lane], h3 = xbuf[192 + lane];
                uint32_t pa = a.x & 1u, pb = b.x & 1u;
                Lbl TG = lxor(lxor(h0, h1), lmask(R, pb));
                Lbl WG = lxor(h0, lmask(TG, pa));
                Lbl TE = lxor(lxor(h2, h3), a);
                Lbl WE = lxor(h2, lmask(lxor(TE, a), pb));
                if (wave == 0) {
                    st_lbl(slot, TG);
                    st_lbl(slot + 64, TE);
                }
                c = lxor(WG, WE);
            }
        } else {
            Lbl h0 = xbuf[lane], h1 = xbuf[64 + lane];
            uint32_t sa = a.x & 1u, sb = b.x & 1u;
            Lbl WG = lxor(h0, lmask(TGe, sa));
            Lbl WE = lxor(h1, lmask(lxor(TEe, a), sb));
            c = lmask(lxor(WG, WE), on ? 1u : 0u);
        }
        return c;
    }
    // two gate steps (gid, gid + 64) at once: 8 (garbler) / 4 (evaluator) hashes over 4 waves
    static __device__ __forceinline__ void and2_quad(TAB lt, Lbl R, W a1, W b1, W a2, W b2, uint64_t gid, Lbl *slot,
                                                     bool on1, bool on2, int wave, Lbl *xbuf, int lane, W &c1, W &c2) {
        const uint64_t gid2 = gid + 64;
        Lbl *slot2 = slot + 128;
        Lbl TG1 = lzero(), TE1 = lzero(), TG2 = lzero(), TE2 = lzero();
        if (!GARBLER) {
            TG1 = ld_lbl(slot); TE1 = ld_lbl(slot + 64);
            TG2 = ld_lbl(slot2); TE2 = ld_lbl(slot2 + 64);
        }
        if (GARBLER) {
            // wave q: hash q of gate 1 and hash q of gate 2 (q = 0: a0, 1: a0^R, 2: b0, 3: b0^R)
            Lbl x[2] = {(wave < 2) ? a1 : b1, (wave < 2) ? a2 : b2};
            if (wave & 1) { x[0] = lxor(x[0], R); x[1] = lxor(x[1], R); }
            uint64_t tw[2] = {2 * gid + (uint64_t)(wave >= 2), 2 * gid2 + (uint64_t)(wave >= 2)};
            Lbl h[2] = {lzero(), lzero()};
            if (on1 || on2) hash_n<2, TAB>(lt, c_aes.rk, x, tw, h, c_aes.rk24);
            xbuf[wave * 64 + lane] = h[0];
            xbuf[256 + wave * 64 + lane] = h[1];
        } else {
            // waves 0,1: gate 1 (a, b); waves 2,3: gate 2 (a, b)
            Lbl x = (wave & 1) ? ((wave < 2) ? b1 : b2) : ((wave < 2) ? a1 : a2);
            uint64_t tw = 2 * ((wave < 2) ? gid : gid2) + (uint64_t)(wave & 1);
            Lbl h = lzero();
            if ((wave < 2) ? on1 : on2) hash_n<1, TAB>(lt, c_aes.rk, &x, &tw, &h, c_aes.rk24);
            xbuf[(wave >> 1) * 256 + (wave & 1) * 64 + lane] = h;
        }
        lds_barrier();
        c1 = lzero();
        c2 = lzero();
        if (GARBLER) {
            if (on1) {
                Lbl h0 = xbuf[lane], h1 = xbuf[64 + lane], h2 = xbuf[128 + lane], h3 = xbuf[192 + lane];
                uint32_t pa = a1.x & 1u, pb = b1.x & 1u;
                Lbl TG = lxor(lxor(h0, h1), lmask(R, pb));
                Lbl WG = lxor(h0, lmask(TG, pa));
                Lbl TE = lxor(lxor(h2, h3), a1);
                Lbl WE = lxor(h2, lmask(lxor(TE, a1), pb));
                if (wave == 0) { st_lbl(slot, TG); st_lbl(slot + 64, TE); }
                c1 = lxor(WG, WE);
            }
            if (on2) {
                Lbl h0 = xbuf[256 + lane], h1 = xbuf[320 + lane], h2 = xbuf[384 + lane], h3 = xbuf[448 + lane];
                uint32_t pa = a2.x & 1u, pb = b2.x & 1u;
                Lbl TG = lxor(lxor(h0, h1), lmask(R, pb));
                Lbl WG = lxor(h0, lmask(TG, pa));
                Lbl TE = lxor(lxor(h2, h3), a2);
                Lbl WE = lxor(h2, lmask(lxor(TE, a2), pb));
                if (wave == 1) { st_lbl(slot2, TG); st_lbl(slot2 + 64, TE); }
                c2 = lxor(WG, WE);
            }
        } else {
            {
                Lbl h0 = xbuf[lane], h1 = xbuf[64 + lane];
                uint32_t sa = a1.x & 1u, sb = b1.x & 1u;
                c1 = lmask(lxor(lxor(h0, lmask(TG1, sa)), lxor(h1, lmask(lxor(TE1, a1), sb))), on1 ? 1u : 0u);
            }
            {
                Lbl h0 = xbuf[256 + lane], h1 = xbuf[320 + lane];
                uint32_t sa = a2.x & 1u, sb = b2.x & 1u;
                c2 = lmask(lxor(lxor(h0, lmask(TG2, sa)), lxor(h1, lmask(lxor(TE2, a2), sb))), on2 ? 1u : 0u);
            }
        }
    }
    __device__ __forceinline__ W load(uint32_t id) const { return ld_lbl(words + (size_t)id * 64 + lane); }
    __device__ __forceinline__ W load2(uint32_t lo, uint32_t hi) const {
        return ld_lbl(words + (size_t)(lane < 32 ? lo : hi) * 64 + (lane & 31));
    }
    __device__ __forceinline__ W load2h(uint32_t lo, uint32_t hi) const {
        return ld_lbl(words + (size_t)(lane < 32 ? lo : hi) * 64 + 32 + (lane & 31));
    }
    __device__ __forceinline__ W load2s(uint32_t lo, uint32_t hi, bool upper) const {
        return ld_lbl(words + (size_t)(lane < 32 ? lo : hi) * 64 + (upper ? 32 : 0) + (lane & 31));
    }
    __device__ __forceinline__ void store2(uint32_t lo, uint32_t hi, W v) {
        if (MODE != MODE_QUAD || wave == 0) st_lbl(words + (size_t)(lane < 32 ? lo : hi) * 64 + (lane & 31), v);
    }
    __device__ __forceinline__ void store(uint32_t id, W v) {
        if (MODE != MODE_QUAD || wave == 0) st_lbl(words + (size_t)id * 64 + lane, v);
    }
    __device__ __forceinline__ void reveal(uint32_t slot, W v) {
        uint64_t m = __ballot(v.x & 1u);
        if (lane == 0 && (MODE != MODE_QUAD || wave == 0)) decode[slot] = m;
    }
};

// MAC launches: one wavefront per record; the TPB/64 waves of a workgroup share the LDS table
// Which record a wave of a MAC workgroup takes next.  A SIMD issues from its OLDEST wave first: of sixteen waves with one
// record each, the four that arrived first on their SIMDs are through after 45 % of the workgroup's time, the last four
// need all of it, and the CU hashes with twelve, eight, four waves meanwhile (profiles/r6_wave_order.txt: the same AES work
// per wave, 19.5 / 28.1 / 35.8 / 43.0 ms by arrival).  So a workgroup owns `per_wg` CONSECUTIVE records (chunk = waves x a
// few) and its waves PULL them from a counter in LDS: all waves stay busy until the chunk is empty, whatever their speed.
// per_wg = waves: the old static assignment (one record per wave), for launches of a few rounds.
#ifndef GC_MAC_WAVE_TRACE
#define GC_MAC_WAVE_TRACE 0     /* profiling builds only: time in the kernel by wave index (scripts/gpu_wave_order.py) */
#endif
#if GC_MAC_WAVE_TRACE
__device__ unsigned long long g_mac_wave_ticks[2 * 16];      // [role][wave index]: summed over workgroups; and counts
__device__ unsigned long long g_mac_wave_count[2 * 16];
#endif
struct MacQueue {
    uint32_t *next;        // LDS
    uint32_t base, end;    // this workgroup's records [base, end)
    __device__ __forceinline__ void init(uint32_t *lds_next, uint32_t per_wg, uint32_t nrec) {
        next = lds_next;
        base = blockIdx.x * per_wg;
        end = base + per_wg < nrec ? base + per_wg : nrec;
        if (threadIdx.x == 0) *lds_next = 0;      // (made visible by the barrier that ends the table fill)
    }
    __device__ __forceinline__ bool pull(uint32_t &wid) {
        uint32_t k = 0;
        if ((threadIdx.x & 63u) == 0) k = atomicAdd(next, 1u);
        wid = base + (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        return wid < end;
    }
};

template <bool GARBLER, int TPB>
__global__ void __launch_bounds__(TPB)
gc_mac_kernel(const Rec *recs, uint32_t nrec, uint32_t per_wg, Lbl *words, Lbl *tab, uint64_t launch_step0, Lbl R, int w, int p) {
    // the four rotated tables: 128 KiB, one workgroup per CU
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    __shared__ uint32_t lds_next;
    MacQueue q;
    q.init(&lds_next, per_wg, nrec);
    lds_tab4_fill(lds_te0);
    const int lane = threadIdx.x & 63;
    typedef GpuBackend<GARBLER, MODE_MAC, LdsTab4> B;
    B be;
    be.R = R;
    be.words = words;
    be.tab = tab;
    be.decode = 0;
    be.xsel = 0;
    be.launch_step0 = launch_step0;
    be.lane = lane;
    be.wave = 0;
    be.xch = 0;
    be.lt = lds_tab4_make(lds_te0);
    typedef Circ<B> C;
    // (A persistent form -- ONE workgroup per CU walking all records -- was measured in round 4: the garbler's launches of a
    // few rounds 12-14 % faster alone, the co-located solve slower, because the evaluator's kernels no longer slip in between
    // the garbler's workgroups.  A chunk of a few records per wave keeps the turnover.)
    uint32_t wid;
    while (q.pull(wid)) {
        Rec r = recs[wid];
        r.cnt = __builtin_amdgcn_readfirstlane(r.cnt);
        r.dst = __builtin_amdgcn_readfirstlane(r.dst);
        r.a = __builtin_amdgcn_readfirstlane(r.a);
        r.b = __builtin_amdgcn_readfirstlane(r.b);
        r.sa = __builtin_amdgcn_readfirstlane(r.sa);
        r.sb = __builtin_amdgcn_readfirstlane(r.sb);
        uint32_t s_lo = __builtin_amdgcn_readfirstlane((uint32_t)r.step0);
        uint32_t s_hi = __builtin_amdgcn_readfirstlane((uint32_t)(r.step0 >> 32));
        be.step = ((uint64_t)s_hi << 32) | s_lo;
        Lbl S = lzero(), Cc = lzero();
        if (__builtin_amdgcn_readfirstlane(r.op) == OP_MAC2) {
            for (uint32_t k = 0; k < r.cnt; k++)
                C::mac2(be, S, Cc, be.load2(r.a + (int32_t)k * r.sa, r.a + (int32_t)(r.cnt + k) * r.sa),
                        be.load2(r.b + (int32_t)k * r.sb, r.b + (int32_t)(r.cnt + k) * r.sb), p);
            be.store2(r.dst, r.dst + 2, S);
            be.store2(r.dst + 1, r.dst + 3, Cc);
            continue;
        }
        for (uint32_t k = 0; k < r.cnt; k++)
            C::mac(be, S, Cc, be.load(r.a + (int32_t)k * r.sa), be.load(r.b + (int32_t)k * r.sb), w, p);
        be.store(r.dst, S);
        be.store(r.dst + 1, Cc);
    }
}

// OP_MACK launches (Karatsuba products, Circ::mack2): the same geometry as gc_mac_kernel, in a kernel of its own.  Four
// inlined gate bodies: the two of the 32 x 32 array loop (one copy: the three arrays of a pair are one loop) and one in
// each of the two recombination loops; the nine sub-product words of a pair wait in scratch memory meanwhile.
// (Out-of-line gates for the recombination, or one kernel for OP_MAC and OP_MACK together, measured no faster / slower:
// the combined kernel spilled 175 VGPRs.)
template <bool GARBLER, int TPB>
__global__ void __launch_bounds__(TPB)
gc_mack_kernel(const Rec *recs, uint32_t nrec, uint32_t per_wg, Lbl *words, Lbl *tab, uint64_t launch_step0, Lbl R, int w, int p) {
#if GC_MAC_WAVE_TRACE
    const unsigned long long t_in = wall_clock64();
#endif
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    __shared__ uint32_t lds_next;
    MacQueue q;
    q.init(&lds_next, per_wg, nrec);
    lds_tab4_fill(lds_te0);
    typedef GpuBackend<GARBLER, MODE_MAC, LdsTab4> B;
    B be;
    be.R = R;
    be.words = words;
    be.tab = tab;
    be.decode = 0;
    be.xsel = 0;
    be.launch_step0 = launch_step0;
    be.lane = threadIdx.x & 63;
    be.wave = 0;
    be.xch = 0;
    be.lt = lds_tab4_make(lds_te0);
    uint32_t wid;
    while (q.pull(wid)) {
        Rec r = recs[wid];
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
        Lbl S = lzero(), Cc = lzero();
        Circ<B>::mack_rec(be, S, Cc, r.a, r.b, r.sa, r.sb, r.cnt, r.c, p);
        be.store(r.dst, S);
        be.store(r.dst + 1, Cc);
    }
#if GC_MAC_WAVE_TRACE
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_mac_wave_ticks[(GARBLER ? 0 : 16) + (threadIdx.x >> 6)], wall_clock64() - t_in);
        atomicAdd(&g_mac_wave_count[(GARBLER ? 0 : 16) + (threadIdx.x >> 6)], 1ull);
    }
#endif
}

// every other record type.  QUAD = true: one 4-wave workgroup per record (narrow, latency-bound
// launches; TPB = 256); QUAD = false: one wave per record, TPB / 64 records per workgroup (wide
// launches).  TABV picks the AES tables: 4 = four rotated tables in 128 KiB (fewest instructions
// per round; one workgroup per CU), 2 = two tables in 64 KiB (two 4-wave workgroups per CU).
template <bool GARBLER, bool QUAD, int TABV, int TPB>
__global__ void __launch_bounds__(TPB)
gc_exec_kernel(const Rec *recs, uint32_t nrec, Lbl *words, Lbl *tab, uint64_t *decode,
               uint64_t launch_step0, Lbl R, int w, int p) {
    typedef TabSel<TABV> TS;
    __shared__ uint32_t lds_te0[TS::kWords];
    __shared__ Lbl lds_xch[QUAD ? 2 * 512 : 1];   // 16 KiB exchange area of the 4-wave steps
    TS::fill(lds_te0);
    const uint32_t wid = QUAD ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // wide: blockDim <= TPB
    if (wid >= nrec) return;
    typedef GpuBackend<GARBLER, QUAD ? MODE_QUAD : MODE_SOLO, typename TS::T> B;
    B be;
    be.R = R;
    be.words = words;
    be.tab = tab;
    be.decode = decode;
    be.launch_step0 = launch_step0;
    be.lane = threadIdx.x & 63;
    be.wave = QUAD ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    be.xch = lds_xch;
    be.xsel = 0;
    be.lt = TS::make(lds_te0);
    Rec r = recs[wid];
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
    exec_record(be, r, w, p);
}

// second pass of critical-path garbling: the stash holds (a0, b0) of every gate step of the launch (two rows
// per step); one wavefront per step turns them into the half-gates ciphertexts (TG, TE) of the launch's table
// (stash == tab: in place).  All four hashes are recomputed here, in throughput mode (16 waves per CU,
// four-table AES).
template <int TPB>
__global__ void __launch_bounds__(TPB)
gc_tabfill_kernel(const Lbl *stash, Lbl *tab, uint32_t nsteps, uint64_t launch_step0, Lbl R) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    const LdsTab4 lt = lds_tab4_make(lds_te0);
    const int lane = threadIdx.x & 63;
    const uint32_t per = TPB / 64;
    for (uint32_t row = blockIdx.x * per + (threadIdx.x >> 6); row < nsteps; row += gridDim.x * per) {
        Lbl *slot = tab + (size_t)row * 128 + lane;
        const Lbl *src = stash + (size_t)row * 128 + lane;
        const Lbl a0 = ld_lbl(src), b0 = ld_lbl(src + 64);
        Lbl TG = lzero(), TE = lzero();
        if ((a0.x | a0.y | a0.z | a0.w | b0.x | b0.y | b0.z | b0.w) != 0u) {
            const uint64_t gid = (launch_step0 + row) * 64 + (uint64_t)lane;
            (void)garble_and(lt, c_aes.rk, R, a0, b0, gid, TG, TE, c_aes.rk24);
        }
        st_lbl_global_nt(slot, TG);
        st_lbl_global_nt(slot + 64, TE);
    }
}

}  // namespace gc
