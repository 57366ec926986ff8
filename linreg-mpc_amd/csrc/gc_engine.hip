// gc_engine.hip -- host engine + C ABI (include/linreg_gc.h) of the MI355X
// garbled-circuit path.  Device code: gc_device.h.  Circuits: gc_circuits.h.
// Program lowering: gc_program.h.  No CPU fallback: without a HIP device every
// compute entry point returns LGC_ENODEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"
#include "hip_scope.h"
#include "gc_device.h"
#include "gc_program.h"
#include "gc_launch.h"

using namespace gc;

static thread_local char g_err[512] = "";
int lgc_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(x)                                                                               \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) return lgc_fail(LGC_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *lgc_last_error(void) { return g_err; }

// LINREG_TRACE=1: where a run spends its start-up.  One line per mark on stderr, "LGCT <tag> <seconds> <what>", the time
// being CLOCK_MONOTONIC so that the marks of all processes of a run -- and of whoever spawned them -- share one base
// (bench.py builds the `timeline` of its phase12 entries from these lines).  Off: one getenv at the first mark.
static char g_trace_tag[32] = "lgc";
static int trace_on() {
    static const int on = getenv("LINREG_TRACE") != NULL;
    return on;
}
extern "C" void lgc_trace_set_tag(const char *tag) {
    if (tag) snprintf(g_trace_tag, sizeof g_trace_tag, "%s", tag);
}
extern "C" void lgc_trace_mark(const char *what) {
    if (!trace_on()) return;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    fprintf(stderr, "LGCT %s %.6f %s\n", g_trace_tag, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, what ? what : "");
}
extern "C" const char *lgc_version(void) { return "linreg-mpc_amd gc engine r6 (gfx950, half-gates, LDS T-table AES, Karatsuba MAC, Sklansky adders, byte table ring, MAC record queue)"; }
extern "C" int lgc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lgc_need_device(int device) {
    static std::once_flag enter;
    std::call_once(enter, [] { lgc_trace_mark("lib: first device call"); });
    int n = lgc_device_count();
    if (n <= 0) return lgc_fail(LGC_ENODEVICE, "no HIP device visible: the garbled-circuit engine has no CPU fallback");
    if (device < 0 || device >= n) return lgc_fail(LGC_EINVAL, "device %d out of range (%d visible)", device, n);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return lgc_fail(LGC_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    static std::once_flag first;
    std::call_once(first, [] {
        lgc_trace_mark("lib: hip runtime up (device count, hipSetDevice)");
        (void)hipFree(0);                                // the device context itself
        lgc_trace_mark("lib: device context up");
    });
    return LGC_OK;
}

// brings the HIP runtime and the device's context up (any thread: both are process-wide); a host calls it from a thread
// at start-up so that the 60-150 ms overlap its own parsing and connecting
extern "C" int lgc_device_warm(int device) {
    int rc = lgc_need_device(device);
    if (rc) return rc;
    // the first dispatch of a process creates its hardware queue and loads the runtime's own fill / copy kernels: ~20 ms
    void *p = 0;
    if (hipMalloc(&p, 4096) == hipSuccess) {
        (void)hipMemset(p, 0, 4096);
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
    }
    lgc_trace_mark("lib: first dispatch done (warm-up thread)");
    return LGC_OK;
}

// A small pool of HIP streams per device.  hipStreamCreate costs ~10 ms (a hardware queue), and the OT sessions of an
// end-to-end run are created on its critical path (after the base OTs): lgc_preload creates streams ahead of time, a session
// takes one from the pool and gives it back when it is destroyed.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> free_;
    hipError_t take(int device, hipStream_t *out) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].first == device) { *out = free_[i].second; free_.erase(free_.begin() + (long)i); return hipSuccess; }
        }
        return hipStreamCreate(out);
    }
    void give(int device, hipStream_t st) {
        std::lock_guard<std::mutex> g(mu);
        if (free_.size() < 16) { free_.emplace_back(device, st); return; }
        (void)hipStreamDestroy(st);
    }
};
static StreamPool &stream_pool() { static StreamPool p; return p; }
hipError_t lgc_stream_take(int device, hipStream_t *out) { return stream_pool().take(device, out); }
void lgc_stream_give(int device, hipStream_t st) { stream_pool().give(device, st); }

hipError_t p1_tu_touch(hipStream_t st);
hipError_t ot_tu_touch(hipStream_t st);
// what & 1: the phase-1 kernels, what & 2: the OT kernels (their code objects are loaded), what & 4: two pooled streams
extern "C" int lgc_preload(int device, int what) {
    int rc = lgc_need_device(device);
    if (rc) return rc;
    if (what & 1) HIPCHK(p1_tu_touch(0));
    if (what & 2) HIPCHK(ot_tu_touch(0));
    if (what & 4) {
        for (int k = 0; k < 2; k++) {
            hipStream_t st;
            HIPCHK(hipStreamCreate(&st));
            stream_pool().give(device, st);
        }
    }
    HIPCHK(hipDeviceSynchronize());
    lgc_trace_mark("lib: kernels preloaded, streams pooled");
    return LGC_OK;
}

// ---------------------------------------------------------------- constants
static AesTables g_tabs;
static bool g_tabs_built = false;
static const AesTables &tables() {
    static std::once_flag once;
    std::call_once(once, [] { aes_build_tables(g_tabs, kFixedKey); g_tabs_built = true; });
    return g_tabs;
}
// Garbler randomness.  The global offset R and the zero-labels of the input wires are AES-128 in counter mode
// KEYED BY THE SEED (its own key schedule, as the OT column PRG and the TI generator do) -- not the public
// fixed-key gate hash evaluated at seed-dependent points, which would let one guess of the seed be tested
// against every observed label at once.  Block = (index lo, index hi, 0, domain): domain 1 = input label of
// (word id, lane), domain 2 = R.
struct SeedKeys { uint32_t rk[44]; };
SeedKeys seed_keys(const Lbl &seed) {
    AesTables t;
    aes_build_tables(t, reinterpret_cast<const uint8_t *>(&seed));
    SeedKeys k;
    memcpy(k.rk, t.rk, sizeof(k.rk));
    return k;
}
Lbl derive_R(const Lbl &seed) {
    SeedKeys k = seed_keys(seed);
    HostTab ht = {tables().te0};
    uint32_t st[1][4] = {{0u, 0u, 0u, 2u}};
    aes_encrypt_n<1, HostTab>(ht, k.rk, st);
    Lbl R = {st[0][0] | 1u, st[0][1], st[0][2], st[0][3]};     // point-and-permute: lsb(R) = 1
    return R;
}

// ------------------------------------------------------------------ kernels
// fresh input labels: zero-label from the seeded PRG, evaluator side gets the
// label of the actual bit (what the OT / direct transfer would deliver)
__global__ void __launch_bounds__(256)
gc_input_kernel(Lbl *wordsG, Lbl *wordsE, const uint64_t *vals, uint32_t base, uint32_t n, Lbl R, SeedKeys keys, int w) {
    __shared__ uint32_t lds_te0[kLdsTabWords];
    __shared__ uint32_t srk[44];
    if (threadIdx.x < 44) srk[threadIdx.x] = keys.rk[threadIdx.x];
    lds_tab_fill(lds_te0);                 // ends with a barrier
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n) return;
    LdsTab lt = lds_tab_make(lds_te0);
    const uint32_t id = base + k;
    const uint64_t idx = (uint64_t)id * 64 + (uint64_t)lane;
    uint32_t st[1][4] = {{(uint32_t)idx, (uint32_t)(idx >> 32), 0u, 1u}};
    aes_encrypt_n<1, LdsTab>(lt, srk, st);
    Lbl z = {st[0][0], st[0][1], st[0][2], st[0][3]};
    if (lane >= w) z = lzero();
    uint32_t bit = vals ? (uint32_t)(vals[k] >> lane) & 1u : 0u;
    if (lane >= w) bit = 0;
    if (wordsG) st_lbl(wordsG + (size_t)id * 64 + lane, z);
    if (wordsE) st_lbl(wordsE + (size_t)id * 64 + lane, lxor(z, lmask(R, bit)));
}

__global__ void __launch_bounds__(1024)
gc_aes_bench_kernel(uint32_t *out, int blocks_per_lane) {
    // the table variant of the MAC kernels (the roof they are priced against)
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    lds_tab4_fill(lds_te0);
    LdsTab4 lt = lds_tab4_make(lds_te0);
    typedef LdsTab4 TabT;
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[4][4];
    for (int b = 0; b < 4; b++) { s[b][0] = gid; s[b][1] = b; s[b][2] = gid * 2654435761u; s[b][3] = 0x9e3779b9u ^ b; }
    for (int i = 0; i < blocks_per_lane; i += 4) aes_encrypt_n<4, TabT>(lt, c_aes.rk, s, c_aes.rk24);
    uint32_t acc = 0;
    for (int b = 0; b < 4; b++) acc ^= s[b][0] ^ s[b][1] ^ s[b][2] ^ s[b][3];
    out[gid] = acc;
}

// known-answer path: even blocks through the MAC kernels' table variant, odd blocks through the
// single-table variant of the generic kernels, so both are pinned by the FIPS-197 vectors
__global__ void __launch_bounds__(256)
gc_aes_encrypt_kernel(const uint4 *in, uint4 *out, uint32_t n) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = i < n ? in[i] : make_uint4(0, 0, 0, 0);
    uint32_t s[1][4] = {{v.x, v.y, v.z, v.w}}, s1[1][4] = {{v.x, v.y, v.z, v.w}};
    lds_tab4_fill(lds_te0);
    LdsTab4 l4 = lds_tab4_make(lds_te0);
    aes_encrypt_n<1, LdsTab4>(l4, c_aes.rk, s, c_aes.rk24);
    __syncthreads();
    lds_tab_fill(lds_te0);
    LdsTab lt = lds_tab_make(lds_te0);
    aes_encrypt_n<1, LdsTab>(lt, c_aes.rk, s1);
    if (i >= n) return;
    out[i] = (i & 1) ? make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]) : make_uint4(s[0][0], s[0][1], s[0][2], s[0][3]);
}

// the gate hash H(x, t) (gc_aes.h) on n labels: what the record kernels compute per half gate; pins the device code to
// the host code and to OpenSSL's AES (tests)
__global__ void __launch_bounds__(256)
gc_gate_hash_kernel(const uint4 *in, const uint64_t *tweak, uint4 *out, uint32_t n) {
    __shared__ uint32_t lds_te0[2 * kLdsTabWords];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = i < n ? in[i] : make_uint4(0, 0, 0, 0);
    Lbl x = {v.x, v.y, v.z, v.w}, h = lzero();
    uint64_t tw = i < n ? tweak[i] : 0;
    lds_tab4_fill(lds_te0);
    LdsTab4 l4 = lds_tab4_make(lds_te0);
    hash_n<1, LdsTab4>(l4, c_aes.rk, &x, &tw, &h, c_aes.rk24);
    if (i < n) out[i] = make_uint4(h.x, h.y, h.z, h.w);
}

// ------------------------------------------------------------------ program
struct lgc_program {
    Program P;
    std::vector<lgc_launch> launch_view;   // filled by lgc_program_launches; lives as long as the program
};

static int check_system(const lgc_system *sys) {
    if (!sys) return lgc_fail(LGC_EINVAL, "null system");
    if (sys->width != 32 && sys->width != 64) return lgc_fail(LGC_EINVAL, "width must be 32 or 64");
    if (sys->precision < 0 || sys->precision >= sys->width)
        return lgc_fail(LGC_EINVAL, "precision must satisfy 0 <= p < width (src/cmd/linreg.c:85-88)");
    if (sys->d < 1 || sys->d > 4096) return lgc_fail(LGC_EINVAL, "d out of range");
    if (sys->nshares < 1) return lgc_fail(LGC_EINVAL, "nshares must be >= 1");
    if (sys->algorithm < 0 || sys->algorithm > LGC_ALG_DIMCHECK) return lgc_fail(LGC_EINVAL, "Algorithm must be cholesky, ldlt, or cgd.");
    if (sys->algorithm == LGC_ALG_DIMCHECK && (sys->d != 1 || sys->nshares != 2 || sys->normalize))
        return lgc_fail(LGC_EINVAL, "the dimension check is a program of its own: d = 1, nshares = 2, normalize = 0");
    if (sys->algorithm == LGC_ALG_CGD && sys->num_iterations < 0) return lgc_fail(LGC_EINVAL, "negative iteration count");
    return LGC_OK;
}

// (fixed_t)(lambda * (1ll << p)) -- src/fixed.c:3-5 via src/linear.oc:52
static uint64_t lambda_to_fixed(double lambda, int p, int w) {
    double t = lambda * (double)(1ll << p);
    if (w == 32) {
        if (!(t > -2147483649.0 && t < 2147483648.0)) return (uint64_t)(uint32_t)INT32_MIN;
        return (uint64_t)(uint32_t)(int32_t)t;
    }
    if (!(t >= -9223372036854775808.0 && t < 9223372036854775808.0)) return (uint64_t)INT64_MIN;
    return (uint64_t)(int64_t)t;
}

static int build(Program &P, const lgc_system *sys, uint64_t cap_steps = 0, size_t merge_hint = 1) {
    if (cap_steps) P.cap_steps = cap_steps;
    P.merge_hint = merge_hint;
    int iters = sys->algorithm == LGC_ALG_CGD ? sys->num_iterations : 0;
    build_program(P, sys->algorithm, sys->d, sys->width, sys->precision, iters, sys->nshares, sys->normalize,
                  lambda_to_fixed(sys->lambda, sys->precision, sys->width), sys->reveal_inputs, sys->trace);
    if (!P.ranges_ok()) return lgc_fail(LGC_EINVAL, "internal: a record of the lowered program lies outside its word file");
    return LGC_OK;
}

extern "C" int lgc_program_build(lgc_program **out, const lgc_system *sys) {
    int rc = check_system(sys);
    if (rc) return rc;
    if (!out) return lgc_fail(LGC_EINVAL, "null out");
    lgc_program *p = new lgc_program();
    rc = build(p->P, sys);
    if (rc) { delete p; return rc; }
    *out = p;
    return LGC_OK;
}
static int check_sweep(const lgc_system *sys, size_t count, const double *lambdas) {
    int rc = check_system(sys);
    if (rc) return rc;
    if (!lambdas) return lgc_fail(LGC_EINVAL, "null lambdas");
    if (count < 1 || count > 4096) return lgc_fail(LGC_EINVAL, "count must be in 1..4096");
    if (!sys->normalize || sys->trace || sys->reveal_inputs)
        return lgc_fail(LGC_EINVAL, "a sweep needs normalize = 1 (lambda enters there), trace = 0, reveal_inputs = 0");
    return LGC_OK;
}
// the merged program of `count` circuits; cap_steps as for build()
static int build_sweep(Program &P, const lgc_system *sys, size_t count, const double *lambdas, size_t first, uint64_t cap_steps = 0) {
    Program base;
    int rcb = build(base, sys, cap_steps ? cap_steps : kSweepCapSteps, count);
    if (rcb) return rcb;
    if ((uint64_t)base.n_words * count >= (1ull << 31)) return lgc_fail(LGC_EINVAL, "sweep too large: %zu circuits x %u words", count, base.n_words);
    std::vector<uint64_t> lf(count);
    for (size_t t = 0; t < count; t++) lf[t] = lambda_to_fixed(lambdas[t], sys->precision, sys->width);
    if (!replicate_program(P, base, count, lf.data(), first))
        return lgc_fail(LGC_EINVAL, "sweep too large: a circuit of %llu gate steps (or circuit index %zu) does not fit the gate-id stride of a sweep",
                        (unsigned long long)(base.total_steps - base.prefix_steps), first + count);
    if (!P.ranges_ok()) return lgc_fail(LGC_EINVAL, "internal: a record of the merged program lies outside its word file");
    return LGC_OK;
}
extern "C" int lgc_program_build_sweep_at(lgc_program **out, const lgc_system *sys, size_t count, const double *lambdas, size_t first) {
    int rc = check_sweep(sys, count, lambdas);
    if (rc) return rc;
    if (!out) return lgc_fail(LGC_EINVAL, "null argument");
    lgc_program *p = new lgc_program();
    rc = build_sweep(p->P, sys, count, lambdas, first);
    if (rc) { delete p; return rc; }
    *out = p;
    return LGC_OK;
}
extern "C" int lgc_program_build_sweep(lgc_program **out, const lgc_system *sys, size_t count, const double *lambdas) {
    return lgc_program_build_sweep_at(out, sys, count, lambdas, 0);
}
extern "C" int lgc_program_ring_plan(const lgc_program *p, size_t ring_bytes, size_t *ring_bytes_out, size_t *offsets, int64_t *wait_for) {
    if (!p || !offsets || !wait_for) return lgc_fail(LGC_EINVAL, "null argument");
    std::vector<size_t> off;
    std::vector<int64_t> wait;
    size_t rb = plan_table_ring(p->P, ring_bytes, off, wait);
    if (ring_bytes_out) *ring_bytes_out = rb;
    for (size_t i = 0; i < off.size(); i++) { offsets[i] = off[i]; wait_for[i] = wait[i]; }
    return LGC_OK;
}
extern "C" void lgc_program_destroy(lgc_program *p) { delete p; }
extern "C" int lgc_program_info_get(const lgc_program *p, lgc_program_info *info) {
    if (!p || !info) return lgc_fail(LGC_EINVAL, "null argument");
    const Program &P = p->P;
    info->n_records = P.recs.size();
    info->n_launches = P.launches.size();
    info->n_words = P.n_words;
    info->n_reveal = P.n_reveal;
    info->in_base = P.in_base;
    info->rv_beta = P.rv_beta;
    info->rv_trace = P.rv_trace;
    info->rv_inputs = P.rv_ab;
    info->total_steps = P.total_steps;
    info->total_gates = P.total_gates;
    info->max_launch_steps = P.max_launch_steps;
    info->replicas = P.replicas;
    info->word_stride = P.word_stride;
    info->reveal_stride = P.reveal_stride;
    info->shared_end = P.shared_end;
    info->prefix_launches = P.prefix_launches;
    info->prefix_steps = P.prefix_steps;
    info->total_xors = P.total_xors;
    return LGC_OK;
}
static_assert(sizeof(lgc_record) == sizeof(Rec), "record layout");
extern "C" const lgc_record *lgc_program_records(const lgc_program *p) {
    return reinterpret_cast<const lgc_record *>(p->P.recs.data());
}
extern "C" const lgc_launch *lgc_program_launches(const lgc_program *cp) {
    lgc_program *p = const_cast<lgc_program *>(cp);
    std::vector<lgc_launch> &g_launch_tmp = p->launch_view;
    g_launch_tmp.resize(p->P.launches.size());
    for (size_t i = 0; i < g_launch_tmp.size(); i++) {
        const Launch &L = p->P.launches[i];
        lgc_launch o = {L.first_rec, L.nrec, L.step0, L.steps, L.gates, L.mac_only ? 1 : 0};
        g_launch_tmp[i] = o;
    }
    return g_launch_tmp.data();
}

// ------------------------------------------------------------------- solver
struct lgc_solver {
    lgc_system sys;
    Program P;
    int device;
    Lbl R, seed;
    Lbl *wordsG, *wordsE, *tab;
    uint64_t *decG, *decE, *vals;
    Rec *recs;
    hipStream_t stream, streamE, streamT;   // garbler chain, evaluator chain, table passes of critical-path launches
    hipEvent_t ev0, ev1, ev_in;
    // garbled-table ring: launch i writes / reads [tab_off[i], tab_off[i] + its table bytes); before
    // overwriting, the garbler waits for the evaluation of launch tab_wait[i] (the newest earlier
    // launch whose region overlaps; the evaluator chain is in order, so older ones are done too)
    size_t ring_bytes;
    size_t tab_alloc_bytes;              // size of the allocation behind `tab` (>= ring_bytes when it came from the cache)
    std::vector<size_t> tab_off;
    std::vector<int64_t> tab_wait;
    std::vector<hipEvent_t> evG, evE;     // per launch: tables written / tables consumed
    std::vector<hipEvent_t> evC;          // per launch: critical-path garbling done (table pass may start)
    std::vector<hipEvent_t> evs;
    std::vector<uint64_t> hG, hE;
    std::vector<double> tG, tE;
    std::vector<hipEvent_t> ev_iter;     // end of each cgd iteration on the evaluator chain
    std::vector<double> t_iter;
    bool have_shares, ran;
    bool prefix_ready;    // sweep: the words of the shared region are in place for both roles (prefix run here, or imported)
    bool prefix_imported = false;
    lgc_stats st;
    lgc_solver() : wordsG(0), wordsE(0), tab(0), decG(0), decE(0), vals(0), recs(0), stream(0), streamE(0), streamT(0), ev0(0), ev1(0), ev_in(0), ring_bytes(0), tab_alloc_bytes(0),
                   have_shares(false), ran(false), prefix_ready(false) { memset(&st, 0, sizeof(st)); }
};

// The table ring is by far the largest allocation of a solver (twice its largest launch: 18 GiB at d=500, 42 GB for a
// block of eight d=100 circuits), and a hipMalloc of that size that follows the hipFree of another one was measured at
// 1.46 s (the solve it belongs to: 1.07 s).  A destroyed solver therefore parks its ring here -- one per device, the
// larger one wins -- and the next solver on that device takes it if it is big enough.  Garbled tables are what the
// evaluator is given anyway: nothing secret stays in the parked buffer.  lgc_release_cached_memory() frees it.
namespace {
struct RingCache {
    std::mutex mu;
    struct Slot { int device; void *ptr; size_t bytes; };
    std::vector<Slot> slots;
    void *take(int device, size_t need, size_t *bytes) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < slots.size(); i++)
            if (slots[i].device == device && slots[i].bytes >= need) {
                void *p = slots[i].ptr;
                *bytes = slots[i].bytes;
                slots.erase(slots.begin() + (long)i);
                return p;
            }
        return 0;
    }
    void park(int device, void *ptr, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < slots.size(); i++)
            if (slots[i].device == device) {
                if (slots[i].bytes >= bytes) { (void)hipFree(ptr); return; }
                (void)hipFree(slots[i].ptr);
                slots[i].ptr = ptr; slots[i].bytes = bytes;
                return;
            }
        Slot n = {device, ptr, bytes};
        slots.push_back(n);
    }
    void release(int device) {      // device < 0: all
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < slots.size();) {
            if (device < 0 || slots[i].device == device) {
                (void)hipSetDevice(slots[i].device);
                (void)hipFree(slots[i].ptr);
                slots.erase(slots.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
};
RingCache &ring_cache() { static RingCache c; return c; }
}  // namespace

extern "C" void lgc_release_cached_memory(void) { ring_cache().release(-1); }

extern "C" void lgc_solver_destroy(lgc_solver *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->wordsG) (void)hipFree(s->wordsG);
    if (s->wordsE) (void)hipFree(s->wordsE);
    if (s->tab) {
        (void)hipDeviceSynchronize();                    // nothing of this solver may still write into the ring
        ring_cache().park(s->device, s->tab, s->tab_alloc_bytes);
    }
    if (s->decG) (void)hipFree(s->decG);
    if (s->decE) (void)hipFree(s->decE);
    if (s->vals) (void)hipFree(s->vals);
    if (s->recs) (void)hipFree(s->recs);
    for (size_t i = 0; i < s->evs.size(); i++) (void)hipEventDestroy(s->evs[i]);
    for (size_t i = 0; i < s->ev_iter.size(); i++) (void)hipEventDestroy(s->ev_iter[i]);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (size_t i = 0; i < s->evG.size(); i++) (void)hipEventDestroy(s->evG[i]);
    for (size_t i = 0; i < s->evE.size(); i++) (void)hipEventDestroy(s->evE[i]);
    for (size_t i = 0; i < s->evC.size(); i++) (void)hipEventDestroy(s->evC[i]);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->streamE) (void)hipStreamDestroy(s->streamE);
    if (s->streamT) (void)hipStreamDestroy(s->streamT);
    delete s;
}

static int solver_create(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16], size_t count,
                         const double *lambdas, size_t first);
extern "C" int lgc_solver_create(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16]) {
    return solver_create(out, device, sys, seed, 1, 0, 0);
}
extern "C" int lgc_solver_create_sweep_at(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16],
                                          size_t count, const double *lambdas, size_t first) {
    int rc = check_sweep(sys, count, lambdas);
    if (rc) return rc;
    return solver_create(out, device, sys, seed, count, lambdas, first);
}
extern "C" int lgc_solver_create_sweep(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16],
                                       size_t count, const double *lambdas) {
    return lgc_solver_create_sweep_at(out, device, sys, seed, count, lambdas, 0);
}
extern "C" size_t lgc_solver_num_circuits(const lgc_solver *s) { return s ? s->P.replicas : 0; }
static int solver_create(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16], size_t count,
                         const double *lambdas, size_t first) {
    int rc = check_system(sys);
    if (rc) return rc;
    if (!out || !seed) return lgc_fail(LGC_EINVAL, "null argument");
    rc = lgc_need_device(device);
    if (rc) return rc;
    lgc_solver *s = new lgc_solver();
    s->sys = *sys;
    s->device = device;
    if (lambdas) {
        rc = build_sweep(s->P, sys, count, lambdas, first);
        if (rc) { delete s; return rc; }
    } else {
        rc = build(s->P, sys);
        if (rc) { delete s; return rc; }
    }
    memcpy(&s->seed, seed, 16);
    s->R = derive_R(s->seed);
    const Program &P = s->P;
    size_t wbytes = (size_t)P.n_words * 64 * sizeof(Lbl);
    size_t nin = P.nshares * (P.T + P.d);
#define TRY(x)                                                                                   \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            lgc_fail(e_ == hipErrorOutOfMemory ? LGC_ENOMEM : LGC_EHIP, "%s: %s", #x, hipGetErrorString(e_)); \
            lgc_solver_destroy(s);                                                               \
            return e_ == hipErrorOutOfMemory ? LGC_ENOMEM : LGC_EHIP;                            \
        }                                                                                        \
    } while (0)
    TRY(hipStreamCreate(&s->stream));
    TRY(hipStreamCreate(&s->streamE));
    TRY(hipStreamCreate(&s->streamT));
    TRY(hipEventCreate(&s->ev0));
    TRY(hipEventCreate(&s->ev1));
    TRY(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
    // Table ring: the largest launch plus up to 8 GiB for the small launches (dividers, merges, reveals) that the garbler
    // runs ahead of the evaluator (plan_table_ring)
    s->ring_bytes = plan_table_ring(P, 0, s->tab_off, s->tab_wait);
    TRY(hipMalloc(&s->wordsG, wbytes));
    TRY(hipMalloc(&s->wordsE, wbytes));
    s->tab = reinterpret_cast<Lbl *>(ring_cache().take(s->device, s->ring_bytes, &s->tab_alloc_bytes));   // parked by a destroyed solver
    if (!s->tab) {
        hipError_t em = hipMalloc(&s->tab, s->ring_bytes);
        if (em == hipErrorOutOfMemory) {                                                  // a smaller parked ring may be in the way
            (void)hipGetLastError();
            ring_cache().release(s->device);
            em = hipMalloc(&s->tab, s->ring_bytes);
        }
        TRY(em);
        s->tab_alloc_bytes = s->ring_bytes;
    }
    TRY(hipMalloc(&s->decG, (P.n_reveal + 1) * sizeof(uint64_t)));
    TRY(hipMalloc(&s->decE, (P.n_reveal + 1) * sizeof(uint64_t)));
    TRY(hipMalloc(&s->vals, nin * sizeof(uint64_t)));
    TRY(hipMalloc(&s->recs, P.recs.size() * sizeof(Rec)));
    TRY(hipMemcpy(s->recs, P.recs.data(), P.recs.size() * sizeof(Rec), hipMemcpyHostToDevice));
#undef TRY
    s->hG.resize(P.n_reveal + 1);
    s->hE.resize(P.n_reveal + 1);
    *out = s;
    return LGC_OK;
}

extern "C" int lgc_solver_set_shares(lgc_solver *s, const uint64_t *shares) {
    if (!s || !shares) return lgc_fail(LGC_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    size_t nin = s->P.nshares * (s->P.T + s->P.d);
    HIPCHK(hipMemcpy(s->vals, shares, nin * sizeof(uint64_t), hipMemcpyHostToDevice));
    s->have_shares = true;
    return LGC_OK;
}

extern "C" int lgc_solver_run(lgc_solver *s, int profile) {
    if (!s) return lgc_fail(LGC_EINVAL, "null solver");
    // input labels come either from the shares on this device or with an imported prefix; an importing rank that runs
    // twice needs a new import (its prefix words were valid for ONE run and it never had the shares)
    if (!s->have_shares && !s->prefix_ready)
        return lgc_fail(LGC_ESTATE, s->prefix_imported ? "the imported prefix has been consumed: import again before the next run"
                                                       : "lgc_solver_set_shares has not been called");
    HIPCHK(hipSetDevice(s->device));
    const Program &P = s->P;
    size_t wbytes = (size_t)P.n_words * 64 * sizeof(Lbl);
    size_t nin = P.nshares * (P.T + P.d);
    const size_t nl = P.launches.size();
    // events: 3 per launch (before garble, between, after evaluate) for MAC launches
    // always; for every launch when profiling
    size_t need_ev = 3 * nl;
    while (s->evs.size() < need_ev) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        s->evs.push_back(e);
    }
    while (s->evG.size() < nl) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        s->evG.push_back(a);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        s->evE.push_back(b);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        s->evC.push_back(b);
    }
    // Garbler chain on `stream`, evaluator chain on `streamE`.  Evaluate(k) waits for garble(k);
    // garble(k) waits for the evaluation of the launch whose ring region it overwrites.  With
    // profile != 0 the two chains are serialised so that per-kernel times are exclusive.
    hipStream_t sG = s->stream, sE = profile ? s->stream : s->streamE;
    while (s->ev_iter.size() < P.iter_launch.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        s->ev_iter.push_back(e);
    }
    size_t next_iter = 0;
    HIPCHK(hipEventRecord(s->ev0, sG));
    // sweep block whose shared prefix is already in place (lgc_solver_prefix_garble / _import): the words of
    // the shared region stay, for both roles; the prefix launches are not run again
    const bool pre = s->prefix_ready && P.prefix_launches > 0;
    const size_t keep = pre ? (size_t)P.shared_end * 64 * sizeof(Lbl) : 0;
    HIPCHK(hipMemsetAsync(reinterpret_cast<char *>(s->wordsG) + keep, 0, wbytes - keep, sG));
    HIPCHK(hipMemsetAsync(reinterpret_cast<char *>(s->wordsE) + keep, 0, wbytes - keep, sG));
    HIPCHK(hipMemsetAsync(s->decG, 0, (P.n_reveal + 1) * sizeof(uint64_t), sG));
    HIPCHK(hipMemsetAsync(s->decE, 0, (P.n_reveal + 1) * sizeof(uint64_t), sG));
    if (!pre) {   // fresh input labels (one set for all circuits of a sweep: they share the prefix)
        dim3 grid((unsigned)((nin + 3) / 4)), block(256);
        hipLaunchKernelGGL(gc_input_kernel, grid, block, 0, sG, s->wordsG, s->wordsE, s->vals, P.in_base, (uint32_t)nin, s->R, seed_keys(s->seed), P.w);
        HIPCHK(hipGetLastError());
    }
    if (!profile) {   // the evaluator chain starts after the input labels are in place
        HIPCHK(hipEventRecord(s->ev_in, sG));
        HIPCHK(hipStreamWaitEvent(sE, s->ev_in, 0));
    }
    for (size_t i = 0; i < nl; i++) {
        const Launch &L = P.launches[i];
        Lbl *tab = reinterpret_cast<Lbl *>(reinterpret_cast<char *>(s->tab) + s->tab_off[i]);
        bool timed = profile || L.mac_only;
        if (!profile && s->tab_wait[i] >= 0) HIPCHK(hipStreamWaitEvent(sG, s->evE[(size_t)s->tab_wait[i]], 0));
        // MAC launches are shaped to fill whole rounds of the chip (gc_program.h: kRoundRecs): a garbler
        // MAC launch sharing the CUs with the previous launch's evaluator would break both into ragged rounds
        if (!profile && L.mac_only && i > 0 && P.launches[i - 1].mac_only && L.nrec >= kExclusiveMac &&
            P.launches[i - 1].nrec >= kExclusiveMac)
            HIPCHK(hipStreamWaitEvent(sG, s->evE[i - 1], 0));
        if (timed) HIPCHK(hipEventRecord(s->evs[3 * i], sG));
        const LaunchMode modeG = gc_launch_mode(L, true);      // read once: record kernel and table pass agree
        if (pre && i < P.prefix_launches) {          // tables already in the ring
            if (timed) HIPCHK(hipEventRecord(s->evs[3 * i + 1], sG));
            if (!profile) HIPCHK(hipEventRecord(s->evG[i], sG));
        } else if (profile || !gc_mode_is_crit(modeG, L)) {
            HIPCHK(gc_launch_records<true>(modeG, s->recs, L, s->wordsG, s->decG, tab, s->R, s->P.w, s->P.p, sG));
            if (gc_mode_is_crit(modeG, L)) HIPCHK(gc_launch_tabfill(L, tab, tab, s->R, sG));
            if (timed) HIPCHK(hipEventRecord(s->evs[3 * i + 1], sG));
            if (!profile) HIPCHK(hipEventRecord(s->evG[i], sG));
        } else {
            // critical path on the garbler chain, table pass on the side stream: only the evaluation waits for it.
            // The stash is the launch's own ring region (in place): this ring is private to the process
            HIPCHK(gc_launch_records<true>(modeG, s->recs, L, s->wordsG, s->decG, tab, s->R, s->P.w, s->P.p, sG));
            if (timed) HIPCHK(hipEventRecord(s->evs[3 * i + 1], sG));
            HIPCHK(hipEventRecord(s->evC[i], sG));
            HIPCHK(hipStreamWaitEvent(s->streamT, s->evC[i], 0));
            HIPCHK(gc_launch_tabfill(L, tab, tab, s->R, s->streamT));
            HIPCHK(hipEventRecord(s->evG[i], s->streamT));
        }
        if (!profile) HIPCHK(hipStreamWaitEvent(sE, s->evG[i], 0));
        const bool done_e = pre && i < P.prefix_launches;                     // evaluated with the prefix
        if (profile) {
            if (!done_e) HIPCHK(gc_launch<false>(s->recs, L, s->wordsE, s->decE, tab, s->R, s->P.w, s->P.p, sE));
            HIPCHK(hipEventRecord(s->evs[3 * i + 2], sE));
        } else {
            if (L.mac_only) HIPCHK(hipEventRecord(s->evs[3 * i + 2], sE));   // start of the evaluate kernel
            if (!done_e) HIPCHK(gc_launch<false>(s->recs, L, s->wordsE, s->decE, tab, s->R, s->P.w, s->P.p, sE));
            HIPCHK(hipEventRecord(s->evE[i], sE));
        }
        while (next_iter < P.iter_launch.size() && P.iter_launch[next_iter] == i)
            HIPCHK(hipEventRecord(s->ev_iter[next_iter++], sE));
    }
    if (!profile) {   // join the evaluator chain back into the main stream
        if (nl > 0) HIPCHK(hipStreamWaitEvent(sG, s->evE[nl - 1], 0));
    }
    HIPCHK(hipMemcpyAsync(s->hG.data(), s->decG, (P.n_reveal + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, sG));
    HIPCHK(hipMemcpyAsync(s->hE.data(), s->decE, (P.n_reveal + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, sG));
    HIPCHK(hipEventRecord(s->ev1, sG));
    HIPCHK(hipStreamSynchronize(sG));
    HIPCHK(hipStreamSynchronize(s->streamE));
    HIPCHK(hipStreamSynchronize(s->streamT));
    HIPCHK(hipGetLastError());
    lgc_stats &st = s->st;
    memset(&st, 0, sizeof(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    st.seconds_total = ms * 1e-3;
    st.and_gates = P.total_gates;
    st.gate_steps = P.total_steps;
    st.table_bytes = P.total_steps * 128 * sizeof(Lbl);
    st.launches = nl;
    s->tG.assign(nl, 0.0);
    s->tE.assign(nl, 0.0);
    for (size_t i = 0; i < nl; i++) {
        const Launch &L = P.launches[i];
        if (!(profile || L.mac_only)) continue;
        float g = 0, e = 0;
        HIPCHK(hipEventElapsedTime(&g, s->evs[3 * i], s->evs[3 * i + 1]));
        if (profile) HIPCHK(hipEventElapsedTime(&e, s->evs[3 * i + 1], s->evs[3 * i + 2]));
        st.seconds_garble += g * 1e-3;
        st.seconds_eval += e * 1e-3;
        s->tG[i] = g * 1e-3;
        s->tE[i] = e * 1e-3;
        if (L.mac_only) {
            st.seconds_mac_garble += g * 1e-3;
            st.seconds_mac_eval += e * 1e-3;
            st.mac_gates += L.gates;
            st.mac_launches++;
        }
    }
    s->t_iter.assign(P.iter_launch.size(), 0.0);
    for (size_t t = 0; t < P.iter_launch.size(); t++) {
        float e = 0;
        HIPCHK(hipEventElapsedTime(&e, s->ev0, s->ev_iter[t]));
        s->t_iter[t] = e * 1e-3;
    }
    s->ran = true;
    s->prefix_ready = false;      // (one run per prefix: an importing rank holds no shares to make it again)
    return LGC_OK;
}

// ---- shared prefix of a sweep block (multi-GPU sweep: garbled on one rank, broadcast to the others)
extern "C" size_t lgc_solver_prefix_bytes(const lgc_solver *s) {
    if (!s) return 0;
    // both roles' words of the shared region AFTER the prefix has run: the prefix is garbled and evaluated once, on the rank
    // that holds the shares, and nobody else needs its tables (rounds 2-5 shipped the tables and let every rank evaluate them:
    // 116 MB for 64 lambdas of d = 100, and 726 MB once the division by the normalizer had joined the prefix; now 32 MB)
    return 2 * (size_t)s->P.shared_end * 64 * sizeof(Lbl);
}
extern "C" int lgc_solver_prefix_garble(lgc_solver *s) {
    if (!s) return lgc_fail(LGC_EINVAL, "null solver");
    if (!s->have_shares) return lgc_fail(LGC_ESTATE, "lgc_solver_set_shares has not been called");
    if (!s->P.prefix_launches) return lgc_fail(LGC_ESTATE, "not a sweep solver: there is no shared prefix");
    HIPCHK(hipSetDevice(s->device));
    const Program &P = s->P;
    const size_t sbytes = (size_t)P.shared_end * 64 * sizeof(Lbl), nin = P.nshares * (P.T + P.d);
    HIPCHK(hipMemsetAsync(s->wordsG, 0, sbytes, s->stream));
    HIPCHK(hipMemsetAsync(s->wordsE, 0, sbytes, s->stream));
    hipLaunchKernelGGL(gc_input_kernel, dim3((unsigned)((nin + 3) / 4)), dim3(256), 0, s->stream, s->wordsG, s->wordsE, s->vals,
                       P.in_base, (uint32_t)nin, s->R, seed_keys(s->seed), P.w);
    HIPCHK(hipGetLastError());
    for (uint32_t i = 0; i < P.prefix_launches; i++) {
        Lbl *tab = reinterpret_cast<Lbl *>(reinterpret_cast<char *>(s->tab) + s->tab_off[i]);
        HIPCHK(gc_launch<true>(s->recs, P.launches[i], s->wordsG, s->decG, tab, s->R, P.w, P.p, s->stream));
        HIPCHK(gc_launch<false>(s->recs, P.launches[i], s->wordsE, s->decE, tab, s->R, P.w, P.p, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->prefix_ready = true;
    return LGC_OK;
}
static int prefix_copy(lgc_solver *s, char *buf, bool out) {
    const Program &P = s->P;
    const size_t sbytes = (size_t)P.shared_end * 64 * sizeof(Lbl);
    auto cp = [&](void *mine, char *theirs, size_t n) {
        return out ? hipMemcpyAsync(theirs, mine, n, hipMemcpyDeviceToDevice, s->stream)
                   : hipMemcpyAsync(mine, theirs, n, hipMemcpyDeviceToDevice, s->stream);
    };
    HIPCHK(cp(s->wordsG, buf, sbytes));
    HIPCHK(cp(s->wordsE, buf + sbytes, sbytes));
    HIPCHK(hipStreamSynchronize(s->stream));
    return LGC_OK;
}
extern "C" int lgc_solver_prefix_export(lgc_solver *s, void *dev_buf) {
    if (!s || !dev_buf) return lgc_fail(LGC_EINVAL, "null argument");
    if (!s->prefix_ready) return lgc_fail(LGC_ESTATE, "the prefix has not been garbled (lgc_solver_prefix_garble)");
    HIPCHK(hipSetDevice(s->device));
    return prefix_copy(s, static_cast<char *>(dev_buf), true);
}
extern "C" int lgc_solver_prefix_import(lgc_solver *s, const void *dev_buf) {
    if (!s || !dev_buf) return lgc_fail(LGC_EINVAL, "null argument");
    if (!s->P.prefix_launches) return lgc_fail(LGC_ESTATE, "not a sweep solver: there is no shared prefix");
    HIPCHK(hipSetDevice(s->device));
    int rc = prefix_copy(s, const_cast<char *>(static_cast<const char *>(dev_buf)), false);
    if (rc) return rc;
    s->prefix_ready = true;       // the evaluator's input labels came with the prefix (s->vals stays unset: have_shares
    s->prefix_imported = true;    // is NOT touched -- a later run without a fresh import fails instead of garbling garbage)
    return LGC_OK;
}

static int64_t decode_word(const lgc_solver *s, uint32_t slot) {
    uint64_t v = s->hG[slot] ^ s->hE[slot];
    if (s->P.w == 32) return (int64_t)(int32_t)(uint32_t)v;
    return (int64_t)v;
}

extern "C" int lgc_solver_get_beta(lgc_solver *s, int64_t *beta) {
    if (!s || !beta) return lgc_fail(LGC_EINVAL, "null argument");
    if (!s->ran) return lgc_fail(LGC_ESTATE, "solver has not run");
    for (uint32_t t = 0; t < s->P.replicas; t++)
        for (size_t i = 0; i < s->P.d; i++)
            beta[(size_t)t * s->P.d + i] = decode_word(s, s->P.rv_beta + t * s->P.reveal_stride + (uint32_t)i);
    return LGC_OK;
}
extern "C" int lgc_solver_get_trace(lgc_solver *s, int64_t *trace) {
    if (!s || !trace) return lgc_fail(LGC_EINVAL, "null argument");
    if (!s->ran) return lgc_fail(LGC_ESTATE, "solver has not run");
    if (s->P.rv_trace == ~0u) return lgc_fail(LGC_ESTATE, "trace was not requested");
    size_t n = (size_t)s->sys.num_iterations * (s->P.d + 4);
    for (size_t i = 0; i < n; i++) trace[i] = decode_word(s, s->P.rv_trace + (uint32_t)i);
    return LGC_OK;
}
extern "C" int lgc_solver_get_inputs(lgc_solver *s, int64_t *ab) {
    if (!s || !ab) return lgc_fail(LGC_EINVAL, "null argument");
    if (!s->ran) return lgc_fail(LGC_ESTATE, "solver has not run");
    if (s->P.rv_ab == ~0u) return lgc_fail(LGC_ESTATE, "input reveal was not requested");
    for (size_t i = 0; i < s->P.T + s->P.d; i++) ab[i] = decode_word(s, s->P.rv_ab + (uint32_t)i);
    return LGC_OK;
}
extern "C" int lgc_solver_get_stats(lgc_solver *s, lgc_stats *st) {
    if (!s || !st) return lgc_fail(LGC_EINVAL, "null argument");
    *st = s->st;
    return LGC_OK;
}

extern "C" int lgc_solver_get_profile(lgc_solver *s, double *garble_s, double *eval_s, size_t n) {
    if (!s || !garble_s || !eval_s) return lgc_fail(LGC_EINVAL, "null argument");
    if (n != s->tG.size()) return lgc_fail(LGC_EINVAL, "n must equal the number of launches (%zu)", s->tG.size());
    for (size_t i = 0; i < n; i++) { garble_s[i] = s->tG[i]; eval_s[i] = s->tE[i]; }
    return LGC_OK;
}

extern "C" int lgc_solver_get_iterations(lgc_solver *s, uint64_t *and_gates, double *seconds, size_t n) {
    if (!s) return lgc_fail(LGC_EINVAL, "null solver");
    if (!s->ran) return lgc_fail(LGC_ESTATE, "solver has not run");
    if (n != s->P.iter_launch.size())
        return lgc_fail(LGC_EINVAL, "n must equal the number of cgd iterations (%zu)", s->P.iter_launch.size());
    for (size_t t = 0; t < n; t++) {
        if (and_gates) and_gates[t] = s->P.iter_gates[t];
        if (seconds) seconds[t] = s->t_iter[t];
    }
    return LGC_OK;
}

extern "C" int lgc_solve(int device, const lgc_system *sys, const uint8_t seed[16], const uint64_t *shares,
                         int64_t *beta, int64_t *trace, lgc_stats *stats) {
    lgc_solver *s = 0;
    int rc = lgc_solver_create(&s, device, sys, seed);
    if (rc) return rc;
    rc = lgc_solver_set_shares(s, shares);
    if (!rc) rc = lgc_solver_run(s, 0);
    if (!rc && beta) rc = lgc_solver_get_beta(s, beta);
    if (!rc && trace && sys->trace && sys->algorithm == LGC_ALG_CGD) rc = lgc_solver_get_trace(s, trace);
    if (!rc && stats) rc = lgc_solver_get_stats(s, stats);
    lgc_solver_destroy(s);
    return rc;
}

extern "C" void *lgc_host_alloc(size_t bytes) {
    void *p = 0;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { lgc_fail(LGC_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return 0; }
    return p;
}
extern "C" void lgc_host_free(void *p) { if (p) (void)hipHostFree(p); }

// AND-gate count of the REFERENCE's circuit for the same solve (Obliv-C + absentminded-crypto-kit, two-party input path of
// bin/test_linear_system): exact polynomial fits to the gate counts in experiments/results/phase2_{32,64}/*.out
// (every d the reference published: 10, 20, 50, 100, 200, 500).  cgd: count after `iterations`
// iterations as the result files list it (the last row equals the total).  No fit exists for ldlt.
extern "C" int lgc_reference_gate_count(int algorithm, int width, size_t d, int iterations, uint64_t *gates) {
    if (!gates) return lgc_fail(LGC_EINVAL, "null argument");
    if ((width != 32 && width != 64) || d < 1 || iterations < 0) return lgc_fail(LGC_EINVAL, "bad argument");
    const unsigned __int128 D = d;
    unsigned __int128 v = 0;
    if (algorithm == LGC_ALG_CGD) {
        unsigned __int128 per, tot20;
        if (width == 64) { per = 19300 * D * D + 221432 * D + 206191; tot20 = 386233 * D * D + 4534169 * D + 4123635; }
        else { per = 4064 * D * D + 29155 * D + 8728; tot20 = (162591 * D * D + 1175767 * D) / 2 + 174560; }
        v = tot20 - 20 * per + (unsigned __int128)iterations * per;
    } else if (algorithm == LGC_ALG_CHOLESKY) {
        if (width == 64) v = (9572 * D * D * D + 3 * 71691 * D * D + 584068 * D) / 3 + 31;
        else v = (4064 * D * D * D + 37899 * D * D + 49489 * D) / 6 + 31;
    } else {
        return lgc_fail(LGC_EINVAL, "the reference published no gate counts for this algorithm");
    }
    *gates = (uint64_t)v;
    return LGC_OK;
}

extern "C" void lgc_set_karatsuba(int on) { program_karatsuba() = on != 0; }
extern "C" void lgc_set_table_ring_slack(size_t bytes) {
    ring_slack_bytes() = bytes ? ((bytes + 4095) & ~(size_t)4095) : kRingSlackBytes;
    ring_cache().release(-1);                 // a parked ring was sized with the old slack
}
extern "C" int lgc_gate_hash_eval(int device, const uint8_t *labels, const uint64_t *tweaks, uint8_t *out, size_t n) {
    if (!labels || !tweaks || !out) return lgc_fail(LGC_EINVAL, "null argument");
    DevFree dev_guard;
    int rc = lgc_need_device(device);
    if (rc) return rc;
    uint4 *di = 0, *dout = 0;
    uint64_t *dt = 0;
    HIPCHK(hipMalloc(&di, n * 16 + 16)); dev_guard.add(di);
    HIPCHK(hipMalloc(&dout, n * 16 + 16)); dev_guard.add(dout);
    HIPCHK(hipMalloc(&dt, n * 8 + 8)); dev_guard.add(dt);
    HIPCHK(hipMemcpy(di, labels, n * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dt, tweaks, n * 8, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(gc_gate_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, di, dt, dout, (uint32_t)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost));
    return LGC_OK;
}
extern "C" void lgc_set_split_kernels(int garbler, int evaluator) {
    gc_split_enabled(true).store(garbler != 0);
    gc_split_enabled(false).store(evaluator != 0);
}

// --------------------------------------------------------- micro-benchmarks
extern "C" int lgc_aes_bench(int device, int waves, int blocks_per_lane, double *rate, uint32_t *check) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    int rc = lgc_need_device(device);
    if (rc) return rc;
    if (waves < 16 || blocks_per_lane < 4) return lgc_fail(LGC_EINVAL, "waves >= 16 and blocks_per_lane >= 4 required");
    blocks_per_lane &= ~3;
    int nblk = waves / 16;
    uint32_t *out = 0;
    HIPCHK(hipMalloc(&out, (size_t)nblk * 1024 * 4)); dev_guard.add(out);
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(gc_aes_bench_kernel, dim3(nblk), dim3(1024), 0, 0, out, 4);   // warm-up
    HIPCHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(gc_aes_bench_kernel, dim3(nblk), dim3(1024), 0, 0, out, blocks_per_lane);
    HIPCHK(hipEventRecord(b, 0));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h((size_t)nblk * 1024);
    HIPCHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    uint32_t c = 0;
    for (size_t i = 0; i < h.size(); i++) c ^= h[i];
    if (check) *check = c;
    if (rate) *rate = (double)nblk * 1024.0 * (double)blocks_per_lane / (ms * 1e-3);

    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return LGC_OK;
}

extern "C" int lgc_aes_encrypt(int device, const uint8_t *in, uint8_t *out, size_t n) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    int rc = lgc_need_device(device);
    if (rc) return rc;
    uint4 *di = 0, *dout = 0;
    HIPCHK(hipMalloc(&di, n * 16)); dev_guard.add(di);
    HIPCHK(hipMalloc(&dout, n * 16)); dev_guard.add(dout);
    HIPCHK(hipMemcpy(di, in, n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gc_aes_encrypt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, di, dout, (uint32_t)n);
    HIPCHK(hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost));

    return LGC_OK;
}
