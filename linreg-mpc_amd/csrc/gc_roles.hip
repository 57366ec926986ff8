// gc_roles.hip -- garbler (CSP, party 1) and evaluator (party 2) as separate objects, for
// deployments where the two run in different processes / on different hosts, as in the
// reference (src/cmd/linreg.c:145-199).  The host moves bytes between them (tables, labels,
// decode bits); nothing here touches a socket.
//
//   garbler   : lgc_party_input_pairs / lgc_party_encode_inputs -> labels for the OT / transfer
//               for k in launches: lgc_party_garble(k) -> table bytes -> (network) ->
//   evaluator : lgc_party_set_input_labels;  lgc_party_evaluate(k, table bytes)
//   end       : garbler lgc_party_decode_bits -> evaluator lgc_party_finish -> beta (revealed to
//               party 2 only, cgd.oc:206-208)
#include <hip/hip_runtime.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <vector>

#include "../../include/linreg_gc.h"
#include "../../include/linreg_gc_sweep.h"
#include "../../include/linreg_gc_debug.h"
#include "hip_scope.h"
#include "gc_device.h"
#include "gc_program.h"
#include "gc_launch.h"

using namespace gc;

#define RCHK(x)                                                                              \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) return lgc_fail(LGC_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum { kPartyEvents = 64 };
struct lgc_party {
    lgc_system sys;
    Program P;
    int device, role;
    Lbl R, seed;
    Lbl *words, *tab;
    uint64_t *dec;
    Rec *recs;
    bool labels_ready;
    std::vector<uint64_t> hdec;
    // device-resident table ring shared between a garbler and an evaluator process on one node
    // (hipIpc: same GPU, or a peer GPU over xGMI): slot k % ring_slots holds launch k's tables
    Lbl *ring;
    int ring_slots;
    size_t ring_slot_bytes;
    bool ring_imported;
    // ... or, as a BYTE ring (lgc_party_ring_create_bytes): launch k owns [ring_off[k], ring_off[k] + its table bytes) of
    // ring_bytes, laid out by plan_table_ring on both sides (same program, same size: same offsets); before the garbler
    // overwrites a range it waits for the evaluation of launch ring_wait[k].  ring_slots == 0 in this mode.
    size_t ring_bytes;
    std::vector<size_t> ring_off;
    std::vector<int64_t> ring_wait;
    size_t tab_bytes;      // size of the private buffer `tab`
    // asynchronous garbling into the ring (lgc_party_garble_ring_begin / _wait): record kernels on the null stream, the table
    // passes of critical-path launches on a stream of their own, two stashes in turn, one completion event per launch
    hipStream_t s_pass;
    hipEvent_t ev_rec[kPartyEvents], ev_done[kPartyEvents];
    bool async_ready;
    bool one_stream;                 // lgc_party_garble_ring_streams(p, 1): table passes on the record kernels' stream
    Lbl *stash2;
    size_t stash2_bytes;
    uint64_t n_crit;       // critical-path launches begun so far (stash = n_crit & 1)
    int64_t stash_user[2]; // the launch whose table pass read stash k last (-1: none)
    std::atomic<int64_t> begun_hi;   // highest launch begun asynchronously (-1: none); written by the enqueuing thread, read by the host's notifier thread
};

// (m0, m1) = (zero label, zero label ^ R) per input bit of one share: yaoKeyNewPair (input.c:94-101)
__global__ void gc_export_pairs_kernel(const Lbl *words, uint32_t base, uint32_t n, Lbl R, int w, const uint64_t *vals,
                                       Lbl *m0, Lbl *m1) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (uint32_t)w) return;
    uint32_t k = t / (uint32_t)w, lane = t % (uint32_t)w;
    Lbl z = ld_lbl(words + (size_t)(base + k) * 64 + lane);
    if (m1) {
        st_lbl(m0 + t, z);
        st_lbl(m1 + t, lxor(z, R));
    } else {   // labels of known values (garbler's own inputs, feedObliv*)
        uint32_t bit = (uint32_t)(vals[k] >> lane) & 1u;
        st_lbl(m0 + t, lxor(z, lmask(R, bit)));
    }
}
__global__ void gc_import_labels_kernel(Lbl *words, uint32_t base, uint32_t n, int w, const Lbl *labels) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 64u) return;
    uint32_t k = t / 64u, lane = t % 64u;
    Lbl v = lane < (uint32_t)w ? ld_lbl(labels + (size_t)k * w + lane) : lzero();
    st_lbl(words + (size_t)(base + k) * 64 + lane, v);
}

extern "C" void lgc_party_destroy(lgc_party *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    if (p->role == LGC_ROLE_GARBLER) {
        // the word file holds every wire's zero-label and the stashes the (a0, b0) of critical-path launches: any two of
        // them with their counterparts give R.  Cleared before the memory goes back to the driver (which hands it to the
        // next allocation of ANY process on this GPU as it is); R and the seed likewise on the host.
        if (p->words) (void)hipMemset(p->words, 0, (size_t)p->P.n_words * 64 * sizeof(Lbl));
        if (p->tab && p->tab_bytes) (void)hipMemset(p->tab, 0, p->tab_bytes);
        if (p->stash2 && p->stash2_bytes) (void)hipMemset(p->stash2, 0, p->stash2_bytes);
        (void)hipDeviceSynchronize();
        volatile uint32_t *w = reinterpret_cast<volatile uint32_t *>(&p->R);
        for (int i = 0; i < 4; i++) w[i] = 0;
        w = reinterpret_cast<volatile uint32_t *>(&p->seed);
        for (int i = 0; i < 4; i++) w[i] = 0;
    }
    if (p->async_ready) {
        for (int i = 0; i < kPartyEvents; i++) { (void)hipEventDestroy(p->ev_rec[i]); (void)hipEventDestroy(p->ev_done[i]); }
        if (p->s_pass) (void)hipStreamDestroy(p->s_pass);
    }
    if (p->words) (void)hipFree(p->words);
    if (p->tab) (void)hipFree(p->tab);
    if (p->stash2) (void)hipFree(p->stash2);
    if (p->dec) (void)hipFree(p->dec);
    if (p->recs) (void)hipFree(p->recs);
    if (p->ring) { if (p->ring_imported) (void)hipIpcCloseMemHandle(p->ring); else (void)hipFree(p->ring); }
    delete p;
}

static int party_create(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                        size_t max_launch_table_bytes, size_t count, const double *lambdas, size_t first);
extern "C" int lgc_party_create(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                                size_t max_launch_table_bytes) {
    return party_create(out, device, sys, role, seed, max_launch_table_bytes, 1, 0, 0);
}
extern "C" int lgc_party_create_sweep_at(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                                         size_t max_launch_table_bytes, size_t count, const double *lambdas, size_t first) {
    int rc = check_sweep(sys, count, lambdas);
    if (rc) return rc;
    return party_create(out, device, sys, role, seed, max_launch_table_bytes, count, lambdas, first);
}
extern "C" int lgc_party_create_sweep(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                                      size_t max_launch_table_bytes, size_t count, const double *lambdas) {
    return lgc_party_create_sweep_at(out, device, sys, role, seed, max_launch_table_bytes, count, lambdas, 0);
}
extern "C" int lgc_devices_preflight(const int *devices, size_t n) {
    if (!devices || !n) return lgc_fail(LGC_EINVAL, "empty device list");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return lgc_fail(LGC_ENODEVICE, "no HIP device visible");
    for (size_t i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= count)
            return lgc_fail(LGC_EINVAL, "device index %d does not exist (%d device%s visible)", devices[i], count, count == 1 ? "" : "s");
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            if (devices[i] == devices[j]) continue;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
            if (e != hipSuccess) return lgc_fail(LGC_EHIP, "hipDeviceCanAccessPeer(%d, %d): %s", devices[i], devices[j], hipGetErrorString(e));
            if (!can)
                return lgc_fail(LGC_EINVAL, "device %d cannot access device %d (no peer path: not in one xGMI hive, or peer access is "
                                            "switched off): the blocks of a sweep share their prefix by peer copy", devices[i], devices[j]);
        }
    return LGC_OK;
}
extern "C" size_t lgc_party_num_circuits(const lgc_party *p) { return p ? p->P.replicas : 0; }
extern "C" size_t lgc_party_prefix_launches(const lgc_party *p) { return p ? p->P.prefix_launches : 0; }
extern "C" uint64_t lgc_party_prefix_and_gates(const lgc_party *p) {
    uint64_t g = 0;
    if (p) for (uint32_t i = 0; i < p->P.prefix_launches; i++) g += p->P.launches[i].gates;
    return g;
}
// Blocks of one sweep on several GPUs of ONE process (bin/linreg --devices): `src` has garbled / evaluated the prefix
// launches; `dst` (same role, same system, same seed, possibly another device) takes the words of the shared region --
// constant zero, inputs, share sums -- over xGMI (hipMemcpyPeer) and then runs launches [prefix_launches, n) only.
extern "C" int lgc_party_share_prefix(lgc_party *dst, const lgc_party *src) {
    if (!dst || !src) return lgc_fail(LGC_EINVAL, "null party");
    const Program &D = dst->P, &S = src->P;
    if (dst->role != src->role || D.shared_end != S.shared_end || D.prefix_launches != S.prefix_launches || !S.prefix_launches ||
        D.w != S.w || D.p != S.p || D.d != S.d || D.T != S.T || D.nshares != S.nshares ||
        D.prefix_steps != S.prefix_steps || D.in_base != S.in_base)
        return lgc_fail(LGC_EINVAL, "the parties are not blocks of the same sweep");
    // the prefix itself: the same records in the same launches (what the source garbled is what the destination's circuits read)
    for (uint32_t li = 0; li < S.prefix_launches; li++) {
        const Launch &a = D.launches[li], &b = S.launches[li];
        if (a.first_rec != b.first_rec || a.nrec != b.nrec || a.step0 != b.step0 || a.steps != b.steps ||
            (a.nrec && memcmp(&D.recs[a.first_rec], &S.recs[b.first_rec], (size_t)a.nrec * sizeof(Rec)) != 0))
            return lgc_fail(LGC_EINVAL, "the parties are not blocks of the same sweep: their prefix launches differ");
    }
    // both blocks hash under one R: their gate steps beyond the prefix must not meet (replicate_program lays the circuits of
    // a sweep on a canonical stride, so this only fails for blocks that cover a common circuit)
    auto body = [](const Program &P, uint64_t &lo, uint64_t &hi) {
        lo = ~0ull; hi = 0;
        for (size_t li = P.prefix_launches; li < P.launches.size(); li++) {
            const Launch &L = P.launches[li];
            if (!L.steps) continue;
            if (L.step0 < lo) lo = L.step0;
            if (L.step0 + L.steps > hi) hi = L.step0 + L.steps;
        }
    };
    uint64_t dlo, dhi, slo, shi;
    body(D, dlo, dhi); body(S, slo, shi);
    if (dlo < shi && slo < dhi) return lgc_fail(LGC_EINVAL, "the blocks overlap in gate steps: they cover a common circuit of the sweep");
    if (dst->role == LGC_ROLE_GARBLER && memcmp(&dst->R, &src->R, sizeof(Lbl)) != 0)
        return lgc_fail(LGC_EINVAL, "garbler blocks of one sweep share the seed");
    if (!src->labels_ready) return lgc_fail(LGC_ESTATE, "the source has no input labels yet");
    RCHK(hipSetDevice(src->device));
    RCHK(hipDeviceSynchronize());                       // the source's prefix launches are complete
    RCHK(hipSetDevice(dst->device));
    const size_t n = (size_t)src->P.shared_end * 64 * sizeof(Lbl);
    if (dst->device == src->device) RCHK(hipMemcpy(dst->words, src->words, n, hipMemcpyDeviceToDevice));
    else RCHK(hipMemcpyPeer(dst->words, dst->device, src->words, src->device, n));
    RCHK(hipDeviceSynchronize());
    dst->labels_ready = true;
    return LGC_OK;
}
static int party_create(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                        size_t max_launch_table_bytes, size_t count, const double *lambdas, size_t first) {
    int rc = check_system(sys);
    if (rc) return rc;
    if (!out) return lgc_fail(LGC_EINVAL, "null out");
    if (role != LGC_ROLE_GARBLER && role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_EINVAL, "role must be 1 (garbler) or 2 (evaluator)");
    if (role == LGC_ROLE_GARBLER && !seed) return lgc_fail(LGC_EINVAL, "the garbler needs a seed");
    rc = lgc_need_device(device);
    if (rc) return rc;
    lgc_party *p = new lgc_party();
    p->sys = *sys; p->device = device; p->role = role;
    p->words = 0; p->tab = 0; p->dec = 0; p->recs = 0; p->labels_ready = false;
    p->ring = 0; p->ring_slots = 0; p->ring_slot_bytes = 0; p->ring_imported = false; p->ring_bytes = 0; p->tab_bytes = 0;
    p->s_pass = 0; p->async_ready = false; p->one_stream = false; p->stash2 = 0; p->stash2_bytes = 0; p->n_crit = 0; p->stash_user[0] = p->stash_user[1] = -1;
    p->begun_hi = -1;
    if (!max_launch_table_bytes) max_launch_table_bytes = (size_t)256 << 20;
    const uint64_t cap = max_launch_table_bytes / 2048 ? max_launch_table_bytes / 2048 : 1;
    if (lambdas) {
        rc = build_sweep(p->P, sys, count, lambdas, first, cap);
        if (rc) { delete p; return rc; }
    } else {
        rc = build(p->P, sys, cap);
        if (rc) { delete p; return rc; }
    }
    lgc_trace_mark("lib: program lowered");
    memset(&p->R, 0, sizeof(Lbl)); memset(&p->seed, 0, sizeof(Lbl));
    if (role == LGC_ROLE_GARBLER) {
        memcpy(&p->seed, seed, 16);
        p->R = derive_R(p->seed);
    }
    const Program &P = p->P;
    size_t wbytes = (size_t)P.n_words * 64 * sizeof(Lbl);
    hipError_t e;
    // (p->tab -- one launch of tables: socket mode, and the garbler's private stash in ring mode -- is allocated on first
    // use: an evaluator that reads its tables from the ring never needs it)
    if ((e = hipMalloc(&p->words, wbytes)) != hipSuccess ||
        (e = hipMalloc(&p->dec, (P.n_reveal + 1) * 8)) != hipSuccess || (e = hipMalloc(&p->recs, P.recs.size() * sizeof(Rec))) != hipSuccess) {
        lgc_party_destroy(p);
        return lgc_fail(LGC_ENOMEM, "hipMalloc: %s", hipGetErrorString(e));
    }
    RCHK(hipMemcpy(p->recs, P.recs.data(), P.recs.size() * sizeof(Rec), hipMemcpyHostToDevice));
    RCHK(hipMemset(p->words, 0, wbytes));
    RCHK(hipMemset(p->dec, 0, (P.n_reveal + 1) * 8));
    if (role == LGC_ROLE_GARBLER) {   // fresh zero-labels for every input word
        size_t nin = P.nshares * (P.T + P.d);
        hipLaunchKernelGGL(gc_input_kernel, dim3((unsigned)((nin + 3) / 4)), dim3(256), 0, 0, p->words, (Lbl *)0,
                           (const uint64_t *)0, P.in_base, (uint32_t)nin, p->R, seed_keys(p->seed), P.w);
        RCHK(hipDeviceSynchronize());
        p->labels_ready = true;
    }
    p->hdec.resize(P.n_reveal + 1);
    // the code objects of this role's record kernels now (creation runs beside phase 1), not inside the first launches
    if (role == LGC_ROLE_GARBLER) RCHK(gc_preload<true>(P.launches, 0));
    else RCHK(gc_preload<false>(P.launches, 0));
    RCHK(hipDeviceSynchronize());
    *out = p;
    lgc_trace_mark(role == LGC_ROLE_GARBLER ? "lib: word file, records, input zero-labels on the device" : "lib: word file and records on the device");
    return LGC_OK;
}

extern "C" size_t lgc_party_num_launches(const lgc_party *p) { return p ? p->P.launches.size() : 0; }
extern "C" size_t lgc_party_table_bytes(const lgc_party *p, size_t launch) {
    return (p && launch < p->P.launches.size()) ? (size_t)p->P.launches[launch].steps * 2048 : 0;
}
extern "C" size_t lgc_party_input_bits(const lgc_party *p) { return p ? (p->P.T + p->P.d) * (size_t)p->P.w : 0; }
extern "C" size_t lgc_party_num_reveal(const lgc_party *p) { return p ? p->P.n_reveal : 0; }
extern "C" uint64_t lgc_party_and_gates(const lgc_party *p) { return p ? p->P.total_gates : 0; }
// 32 bytes that two parties compare before the first table moves: everything the two roles of a solve must agree on -- the
// records (operands, constants such as lambda, gate-step numbers), the launch boundaries, width, precision, gate hash.
// Four multiplicative word hashes; a check against MISCONFIGURATION (a flag given to one party only), not against a cheating peer.
extern "C" int lgc_party_program_fingerprint(const lgc_party *p, uint8_t out[32]) {
    if (!p || !out) return lgc_fail(LGC_EINVAL, "null argument");
    const Program &P = p->P;
    uint64_t h[4] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull, 0x9e3779b97f4a7c15ull, 0xd6e8feb86659fd93ull};
    const uint64_t mul[4] = {0x100000001b3ull, 0xff51afd7ed558ccdull, 0xc4ceb9fe1a85ec53ull, 0x9fb21c651e98df25ull};
    auto mix = [&](uint64_t v) {
        for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ v) * mul[k]; h[k] ^= h[k] >> 29; }
    };
    const uint64_t head[] = {(uint64_t)P.w, (uint64_t)P.p, (uint64_t)P.d, (uint64_t)P.nshares, (uint64_t)P.n_words, P.n_reveal,
                             P.in_base, P.rv_beta, P.replicas, P.shared_end, P.prefix_launches, P.total_steps, P.total_gates,
                             (uint64_t)P.recs.size(), (uint64_t)P.launches.size()};
    for (uint64_t v : head) mix(v);
    static_assert(sizeof(Rec) % 8 == 0, "records are hashed as 64-bit words");
    const uint64_t *w = reinterpret_cast<const uint64_t *>(P.recs.data());
    for (size_t i = 0, n = P.recs.size() * (sizeof(Rec) / 8); i < n; i++) mix(w[i]);
    for (const Launch &L : P.launches) { mix(((uint64_t)L.first_rec << 32) | L.nrec); mix(L.step0); mix(L.steps); }
    memcpy(out, h, 32);
    return LGC_OK;
}
extern "C" int lgc_party_iteration_marks(const lgc_party *p, uint32_t *launch, uint64_t *and_gates, size_t n) {
    if (!p) return lgc_fail(LGC_EINVAL, "null party");
    if (n != p->P.iter_launch.size())
        return lgc_fail(LGC_EINVAL, "n must equal the number of cgd iterations (%zu)", p->P.iter_launch.size());
    for (size_t t = 0; t < n; t++) {
        if (launch) launch[t] = p->P.iter_launch[t];
        if (and_gates) and_gates[t] = p->P.iter_gates[t];
    }
    return LGC_OK;
}

static int export_labels(lgc_party *p, size_t share, const uint64_t *values, uint8_t *m0, uint8_t *m1) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "only the garbler owns label pairs");
    if (share >= p->P.nshares) return lgc_fail(LGC_EINVAL, "share index out of range");
    RCHK(hipSetDevice(p->device));
    const uint32_t n = (uint32_t)(p->P.T + p->P.d);
    const size_t bits = (size_t)n * p->P.w;
    Lbl *d0 = 0, *d1 = 0; uint64_t *dv = 0;
    // both labels of every input bit pass through these two buffers (their XOR is R): cleared before they are freed
    RCHK(hipMalloc(&d0, bits * 16)); dev_guard.add_secret(d0, bits * 16);
    if (m1) RCHK(hipMalloc(&d1, bits * 16));
    dev_guard.add_secret(d1, bits * 16);
    if (values) { RCHK(hipMalloc(&dv, n * 8)); dev_guard.add(dv); RCHK(hipMemcpy(dv, values, n * 8, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(gc_export_pairs_kernel, dim3((unsigned)((bits + 255) / 256)), dim3(256), 0, 0, p->words,
                       p->P.in_base + (uint32_t)(share * n), n, p->R, p->P.w, dv, d0, d1);
    RCHK(hipGetLastError());
    RCHK(hipMemcpy(m0, d0, bits * 16, hipMemcpyDeviceToHost));
    if (m1) RCHK(hipMemcpy(m1, d1, bits * 16, hipMemcpyDeviceToHost));

    return LGC_OK;
}
// the same with the pairs left in device memory of the caller (two buffers of input_bits x 16 bytes on the party's device):
// the label OT of a provider then reads them where they are (lgc_ot_sender_set_device_io) -- both labels of every input bit
// never pass through the host.  The caller owns the buffers and clears them before freeing (lgc_dev_free_secret).
extern "C" int lgc_party_input_pairs_dev(lgc_party *p, size_t share, void *d0, void *d1) {
    if (!p || !d0 || !d1) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "only the garbler owns label pairs");
    if (share >= p->P.nshares) return lgc_fail(LGC_EINVAL, "share index out of range");
    RCHK(hipSetDevice(p->device));
    const uint32_t n = (uint32_t)(p->P.T + p->P.d);
    const size_t bits = (size_t)n * p->P.w;
    hipLaunchKernelGGL(gc_export_pairs_kernel, dim3((unsigned)((bits + 255) / 256)), dim3(256), 0, 0, p->words,
                       p->P.in_base + (uint32_t)(share * n), n, p->R, p->P.w, (const uint64_t *)0, (Lbl *)d0, (Lbl *)d1);
    RCHK(hipGetLastError());
    RCHK(hipDeviceSynchronize());
    return LGC_OK;
}
extern "C" int lgc_party_input_pairs(lgc_party *p, size_t share, uint8_t *m0, uint8_t *m1) {
    if (!p || !m0 || !m1) return lgc_fail(LGC_EINVAL, "null argument");
    return export_labels(p, share, 0, m0, m1);
}
extern "C" int lgc_party_encode_inputs(lgc_party *p, size_t share, const uint64_t *values, uint8_t *labels_out) {
    if (!p || !values || !labels_out) return lgc_fail(LGC_EINVAL, "null argument");
    return export_labels(p, share, values, labels_out, 0);
}
// the same from device memory (input_bits x 16 bytes on the party's device, or a peer-mapped buffer: bin/linreg --input_ring)
extern "C" int lgc_party_set_input_labels_dev(lgc_party *p, size_t share, const void *dev_labels) {
    if (!p || !dev_labels) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "only the evaluator imports labels");
    if (share >= p->P.nshares) return lgc_fail(LGC_EINVAL, "share index out of range");
    RCHK(hipSetDevice(p->device));
    const uint32_t n = (uint32_t)(p->P.T + p->P.d);
    hipLaunchKernelGGL(gc_import_labels_kernel, dim3((unsigned)((n * 64u + 255) / 256)), dim3(256), 0, 0, p->words,
                       p->P.in_base + (uint32_t)(share * n), n, p->P.w, (const Lbl *)dev_labels);
    RCHK(hipGetLastError());
    RCHK(hipDeviceSynchronize());
    p->labels_ready = true;
    return LGC_OK;
}
extern "C" int lgc_party_set_input_labels(lgc_party *p, size_t share, const uint8_t *labels) {
    DevFree dev_guard;   // temporary device buffers are released on every return path
    if (!p || !labels) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "only the evaluator imports labels");
    if (share >= p->P.nshares) return lgc_fail(LGC_EINVAL, "share index out of range");
    RCHK(hipSetDevice(p->device));
    const uint32_t n = (uint32_t)(p->P.T + p->P.d);
    const size_t bits = (size_t)n * p->P.w;
    Lbl *d = 0;
    RCHK(hipMalloc(&d, bits * 16)); dev_guard.add(d);
    RCHK(hipMemcpy(d, labels, bits * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gc_import_labels_kernel, dim3((unsigned)((n * 64u + 255) / 256)), dim3(256), 0, 0, p->words,
                       p->P.in_base + (uint32_t)(share * n), n, p->P.w, d);
    RCHK(hipGetLastError());
    RCHK(hipDeviceSynchronize());

    p->labels_ready = true;
    return LGC_OK;
}

// tab: where the launch's garbled tables go / come from (0: the party's private buffer).  A garbler that writes into
// memory the evaluator process maps (the hipIpc ring) keeps the zero-label stash of critical-path launches in its
// PRIVATE buffer p->tab: only finished ciphertexts are ever stored to the shared slot (src/input.c:94-108 -- label
// pairs never leave the CSP)
// Bytes of stash a garbler needs when its tables go to a ring: the largest launch that CAN run as critical-path garbling
// (few records: gc_launch_mode; the run-time switches only ever turn such launches into ordinary ones).  Rounds 2-3 sized the
// private buffer for the largest launch of all -- 2.6 GB at d = 100, allocated inside the first garbling call.
static size_t party_stash_bytes(const lgc_party *p) {
    uint64_t steps = 0;
    const uint32_t lim = kSplitMaxRecs;
    for (const Launch &L : p->P.launches)
        if (L.nrec <= lim && L.steps > steps) steps = L.steps;
    return (size_t)steps * 2048 + 16;
}
static hipError_t party_need_tab(lgc_party *p, size_t bytes) {
    if (p->tab && p->tab_bytes >= bytes) return hipSuccess;
    if (p->tab) { (void)hipDeviceSynchronize(); (void)hipFree(p->tab); p->tab = 0; p->tab_bytes = 0; }
    hipError_t e = hipMalloc(&p->tab, bytes);
    if (e == hipSuccess) p->tab_bytes = bytes;
    return e;
}
static hipError_t party_need_tab(lgc_party *p) { return party_need_tab(p, (size_t)p->P.max_launch_steps * 2048 + 16); }
template <bool G>
static hipError_t party_launch(lgc_party *p, const Launch &L, Lbl *tab = 0, int stages = 3, bool *was_crit = 0) {
    if (!tab) { hipError_t e = party_need_tab(p); if (e != hipSuccess) return e; }
    else if (G) { hipError_t e = party_need_tab(p, party_stash_bytes(p)); if (e != hipSuccess) return e; }
    return gc_launch<G>(p->recs, L, p->words, p->dec, tab ? tab : p->tab, p->R, p->P.w, p->P.p, 0, (G && tab) ? p->tab : (Lbl *)0,
                        stages, was_crit);
}

extern "C" int lgc_party_garble(lgc_party *p, size_t launch, uint8_t *tables_out) {
    if (!p || (!tables_out && lgc_party_table_bytes(p, launch))) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "not the garbler");
    if (launch >= p->P.launches.size()) return lgc_fail(LGC_EINVAL, "launch out of range");
    RCHK(hipSetDevice(p->device));
    const Launch &L = p->P.launches[launch];
    RCHK(party_need_tab(p));
    if (L.steps) RCHK(hipMemset(p->tab, 0, (size_t)L.steps * 2048));   // inactive lanes: defined bytes on the wire
    RCHK(party_launch<true>(p, L));
    if (L.steps) RCHK(hipMemcpy(tables_out, p->tab, (size_t)L.steps * 2048, hipMemcpyDeviceToHost));
    else RCHK(hipDeviceSynchronize());
    return LGC_OK;
}
extern "C" int lgc_party_evaluate(lgc_party *p, size_t launch, const uint8_t *tables_in) {
    if (!p || (!tables_in && lgc_party_table_bytes(p, launch))) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "not the evaluator");
    if (!p->labels_ready) return lgc_fail(LGC_ESTATE, "input labels have not been set");
    if (launch >= p->P.launches.size()) return lgc_fail(LGC_EINVAL, "launch out of range");
    RCHK(hipSetDevice(p->device));
    const Launch &L = p->P.launches[launch];
    RCHK(party_need_tab(p));
    if (L.steps) RCHK(hipMemcpy(p->tab, tables_in, (size_t)L.steps * 2048, hipMemcpyHostToDevice));
    RCHK(party_launch<false>(p, L));
    RCHK(hipDeviceSynchronize());
    return LGC_OK;
}
// ---- table ring (device-resident hand-off; replaces the osend/orecv byte stream of the Yao
// protocol when both roles run on one node)
// hipIpcGetMemHandle / hipIpcOpenMemHandle from two threads of one process at once fail now and then with "invalid argument"
// (seen once bin/linreg built its garbler -- table ring -- on a thread beside the trusted initializer's per-provider rings):
// one IPC call at a time per process
static std::mutex &ipc_mutex() { static std::mutex m; return m; }
// ... and hipIpcGetMemHandle itself fails transiently ("invalid argument", about once in a hundred five-process runs) while
// another thread of the process allocates or launches: tried again a few times before it counts
static hipError_t ipc_get_handle(hipIpcMemHandle_t *h, void *ptr) {
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 8; attempt++) {
        e = hipIpcGetMemHandle(h, ptr);
        if (e == hipSuccess) {
            if (attempt) fprintf(stderr, "linreg_gc: hipIpcGetMemHandle succeeded at attempt %d (transient 'invalid argument' before)\n", attempt + 1);
            return e;
        }
        (void)hipGetLastError();
        struct timespec ts = {0, 2000000};
        nanosleep(&ts, 0);
    }
    return e;
}
static int ring_alloc_export(lgc_party *p, size_t bytes, uint8_t handle_out[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    RCHK(hipSetDevice(p->device));
    hipError_t e = hipMalloc(&p->ring, bytes);
    if (e != hipSuccess) return lgc_fail(LGC_ENOMEM, "hipMalloc(table ring, %zu bytes): %s", bytes, hipGetErrorString(e));
    // the evaluator process maps the whole ring: it must never see what this allocation held before
    // (inactive lanes and padding are not written by the kernels)
    e = hipMemset(p->ring, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(p->ring); p->ring = 0; return lgc_fail(LGC_EHIP, "hipMemset(table ring): %s", hipGetErrorString(e)); }
    hipIpcMemHandle_t h;
    {   // (the lock covers the IPC call only: the allocation and the fill of a multi-GB ring above would serialise every
        // other thread's IPC traffic -- the CSP's per-provider input rings -- behind them)
        std::lock_guard<std::mutex> ipc_lock(ipc_mutex());
        e = ipc_get_handle(&h, p->ring);
    }
    if (e != hipSuccess) {
        (void)hipFree(p->ring); p->ring = 0;
        return lgc_fail(LGC_EHIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
    }
    memcpy(handle_out, &h, 64);
    // the stash of the critical-path launches now as well, not inside the first garbling call
    e = party_need_tab(p, party_stash_bytes(p));
    if (e != hipSuccess) { (void)hipFree(p->ring); p->ring = 0; return lgc_fail(LGC_ENOMEM, "hipMalloc(stash): %s", hipGetErrorString(e)); }
    p->ring_imported = false;
    lgc_trace_mark("lib: table ring allocated, zero-filled, exported");
    return LGC_OK;
}
static int ring_map(lgc_party *p, const uint8_t handle[64]) {
    std::lock_guard<std::mutex> ipc_lock(ipc_mutex());
    RCHK(hipSetDevice(p->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void *ptr = 0;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return lgc_fail(LGC_EHIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    p->ring = (Lbl *)ptr; p->ring_imported = true;
    lgc_trace_mark("lib: table ring mapped");
    return LGC_OK;
}
extern "C" int lgc_party_ring_create(lgc_party *p, int nslots, uint8_t handle_out[64], size_t *slot_bytes) {
    if (!p || !handle_out || !slot_bytes) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "the garbler owns the table ring");
    if (nslots < 1 || nslots > 64) return lgc_fail(LGC_EINVAL, "nslots must be in 1..64");
    if (p->ring) return lgc_fail(LGC_ESTATE, "the ring already exists");
    size_t sb = ((size_t)p->P.max_launch_steps * 2048 + 4095) & ~(size_t)4095;
    if (!sb) sb = 4096;
    int rc = ring_alloc_export(p, sb * (size_t)nslots, handle_out);
    if (rc) return rc;
    p->ring_slots = nslots; p->ring_slot_bytes = sb; p->ring_bytes = 0;
    *slot_bytes = sb;
    return LGC_OK;
}
extern "C" int lgc_party_ring_open(lgc_party *p, const uint8_t handle[64], int nslots, size_t slot_bytes) {
    if (!p || !handle) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "the evaluator opens the garbler's ring");
    if (nslots < 1 || nslots > 64) return lgc_fail(LGC_EINVAL, "nslots must be in 1..64");
    if (slot_bytes < (size_t)p->P.max_launch_steps * 2048) return lgc_fail(LGC_EINVAL, "ring slots are smaller than the largest launch");
    if (p->ring) return lgc_fail(LGC_ESTATE, "the ring is already open");
    int rc = ring_map(p, handle);
    if (rc) return rc;
    p->ring_slots = nslots; p->ring_slot_bytes = slot_bytes; p->ring_bytes = 0;
    return LGC_OK;
}
// The byte ring: the largest launch plus room for what the garbler runs ahead by (default: half as much again, between
// 64 MiB and 4 GiB), instead of `nslots` times the largest launch -- 3.9 GB against 10.5 GB at d = 100, 12 GB against 33 GB
// for config 4: a fresh device allocation costs 30-70 ms per GB on this driver (and the same again when it is released), which
// was the largest single item of an end-to-end run of config 3.  Both processes lay the launches out with plan_table_ring.
static size_t default_ring_bytes(const Program &P) {
    const size_t largest = ((size_t)P.max_launch_steps * 2048 + 4095) & ~(size_t)4095;
    size_t slack = largest / 2;
    if (slack < ((size_t)64 << 20)) slack = (size_t)64 << 20;
    if (slack > ((size_t)4 << 30)) slack = (size_t)4 << 30;
    return largest + slack + 4096;
}
extern "C" int lgc_party_ring_create_bytes(lgc_party *p, size_t ring_bytes, uint8_t handle_out[64], size_t *ring_bytes_out) {
    if (!p || !handle_out || !ring_bytes_out) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "the garbler owns the table ring");
    if (p->ring) return lgc_fail(LGC_ESTATE, "the ring already exists");
    const size_t bytes = plan_table_ring(p->P, ring_bytes ? ring_bytes : default_ring_bytes(p->P), p->ring_off, p->ring_wait);
    int rc = ring_alloc_export(p, bytes, handle_out);
    if (rc) return rc;
    p->ring_slots = 0; p->ring_slot_bytes = 0; p->ring_bytes = bytes;
    *ring_bytes_out = bytes;
    return LGC_OK;
}
extern "C" int lgc_party_ring_open_bytes(lgc_party *p, const uint8_t handle[64], size_t ring_bytes) {
    if (!p || !handle) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "the evaluator opens the garbler's ring");
    if (p->ring) return lgc_fail(LGC_ESTATE, "the ring is already open");
    if (ring_bytes < (size_t)p->P.max_launch_steps * 2048) return lgc_fail(LGC_EINVAL, "the ring is smaller than the largest launch");
    if (plan_table_ring(p->P, ring_bytes, p->ring_off, p->ring_wait) != ring_bytes)
        return lgc_fail(LGC_EINVAL, "the ring does not fit this program");
    int rc = ring_map(p, handle);
    if (rc) return rc;
    p->ring_slots = 0; p->ring_slot_bytes = 0; p->ring_bytes = ring_bytes;
    return LGC_OK;
}
extern "C" int64_t lgc_party_ring_wait_for(const lgc_party *p, size_t launch) {
    if (!p || !p->ring_bytes || launch >= p->ring_wait.size()) return -1;
    return p->ring_wait[launch];
}
static Lbl *ring_slot(lgc_party *p, size_t launch) {
    if (p->ring_bytes) return (Lbl *)((char *)p->ring + p->ring_off[launch]);
    return (Lbl *)((char *)p->ring + (launch % (size_t)p->ring_slots) * p->ring_slot_bytes);
}
extern "C" int lgc_party_garble_ring(lgc_party *p, size_t launch) {
    if (!p) return lgc_fail(LGC_EINVAL, "null party");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "not the garbler");
    if (!p->ring) return lgc_fail(LGC_ESTATE, "lgc_party_ring_create has not been called");
    if (launch >= p->P.launches.size()) return lgc_fail(LGC_EINVAL, "launch out of range");
    RCHK(hipSetDevice(p->device));
    RCHK(party_launch<true>(p, p->P.launches[launch], ring_slot(p, launch)));
    RCHK(hipDeviceSynchronize());      // kernel end = release: the tables are visible to the peer process
    return LGC_OK;
}
// The same in two halves, so that a garbler process need not stop after every launch (profiles/r5_timeline_d100_two_process.txt:
// with a device synchronisation and a token per launch the CSP's chain was its kernels PLUS their table passes PLUS ~22 us of
// host time per launch -- 175 ms at d = 100 CGD-15 against 137 ms for the same kernels in the co-located solver, where the
// table passes run on a third stream and nothing waits for the host):
//   _begin(i)  enqueues launch i -- the record kernel on the null stream, behind launch i - 1; the table pass of a
//              critical-path launch on a stream of its own, behind its record kernel -- and returns at once;
//   _wait(i)   returns once the tables of launch i are complete in the ring (then the peer may be told).
// The caller keeps the ring discipline (lgc_party_ring_wait_for before _begin) and waits for the launches in order; at most
// kPartyEvents - 1 launches may be begun and not yet waited for.  The zero-label stash of critical-path launches alternates
// between two private buffers: the record kernel of launch i + 1 runs beside the table pass of launch i.
static int party_async_setup(lgc_party *p) {
    if (p->async_ready) return LGC_OK;
    if (!p->one_stream) RCHK(hipStreamCreateWithFlags(&p->s_pass, hipStreamNonBlocking));
    for (int i = 0; i < kPartyEvents; i++) {
        RCHK(hipEventCreateWithFlags(&p->ev_rec[i], hipEventDisableTiming));
        RCHK(hipEventCreateWithFlags(&p->ev_done[i], hipEventDisableTiming));
    }
    const size_t sb = party_stash_bytes(p);
    RCHK(party_need_tab(p, sb));
    if (!p->one_stream) {            // (one stream: every kernel in order, one stash)
        hipError_t e = hipMalloc(&p->stash2, sb);
        if (e != hipSuccess) return lgc_fail(LGC_ENOMEM, "hipMalloc(second stash, %zu bytes): %s", sb, hipGetErrorString(e));
        p->stash2_bytes = sb;
    }
    p->async_ready = true;
    return LGC_OK;
}
// streams == 1: the table pass of a critical-path launch stays on the record kernels' stream (no second hardware queue to
// create -- ~10 ms, and more when the process exits --, no second stash): what is left of the asynchronous path is that the
// garbler's stream never waits for the host between launches.  streams == 2 (default): as described above.  Before the first
// lgc_party_garble_ring_begin.
extern "C" int lgc_party_garble_ring_streams(lgc_party *p, int streams) {
    if (!p || (streams != 1 && streams != 2)) return lgc_fail(LGC_EINVAL, "bad argument");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "not the garbler");
    if (p->async_ready) return lgc_fail(LGC_ESTATE, "launches have been begun already");
    p->one_stream = streams == 1;
    return LGC_OK;
}
extern "C" int lgc_party_garble_ring_begin(lgc_party *p, size_t launch) {
    if (!p) return lgc_fail(LGC_EINVAL, "null party");
    if (p->role != LGC_ROLE_GARBLER) return lgc_fail(LGC_ESTATE, "not the garbler");
    if (!p->ring) return lgc_fail(LGC_ESTATE, "lgc_party_ring_create has not been called");
    if (launch >= p->P.launches.size()) return lgc_fail(LGC_EINVAL, "launch out of range");
    RCHK(hipSetDevice(p->device));
    int rc = party_async_setup(p);
    if (rc) return rc;
    const int slot = (int)(launch % kPartyEvents);
    // the events of this slot belong to launch - kPartyEvents until that launch is through
    if (launch >= (size_t)kPartyEvents) RCHK(hipEventSynchronize(p->ev_done[slot]));
    const Launch &L = p->P.launches[launch];
    Lbl *tab = ring_slot(p, launch);
    const int k = (int)(p->n_crit & 1);
    Lbl *stash = (k && p->stash2) ? p->stash2 : p->tab;
    // the launch's mode is decided ONCE (it reads a run-time switch): record kernel and table pass cannot disagree
    const LaunchMode m = gc_launch_mode(L, true);
    const bool crit = gc_mode_is_crit(m, L);
    if (!crit) {
        RCHK(gc_launch_records<true>(m, p->recs, L, p->words, p->dec, tab, p->R, p->P.w, p->P.p, 0));
        RCHK(hipEventRecord(p->ev_done[slot], 0));
    } else {
        // this stash was last read by the table pass of an earlier launch: the record kernel must not overwrite it before
        if (p->stash_user[k] >= 0) RCHK(hipStreamWaitEvent(0, p->ev_done[p->stash_user[k] % kPartyEvents], 0));
        RCHK(gc_launch_records<true>(m, p->recs, L, p->words, p->dec, stash, p->R, p->P.w, p->P.p, 0));
        RCHK(hipEventRecord(p->ev_rec[slot], 0));
        RCHK(hipStreamWaitEvent(p->s_pass, p->ev_rec[slot], 0));
        RCHK(gc_launch_tabfill(L, stash, tab, p->R, p->s_pass));
        RCHK(hipEventRecord(p->ev_done[slot], p->s_pass));
        p->stash_user[k] = (int64_t)launch;
        p->n_crit++;
    }
    if ((int64_t)launch > p->begun_hi.load(std::memory_order_relaxed)) p->begun_hi.store((int64_t)launch, std::memory_order_release);
    return LGC_OK;
}
extern "C" int lgc_party_garble_ring_wait(lgc_party *p, size_t launch) {
    if (!p || !p->async_ready) return lgc_fail(LGC_ESTATE, "nothing was begun (lgc_party_garble_ring_begin)");
    const int64_t hi = p->begun_hi.load(std::memory_order_acquire);
    if ((int64_t)launch > hi) return lgc_fail(LGC_EINVAL, "launch %zu has not been begun", launch);
    // Its event slot has been handed to launch + kPartyEvents already: _begin synchronised on this launch's completion event
    // before it reused the slot, so the launch is complete.  (The enqueuing thread is throttled by the GPU, not by the
    // thread that waits: after a long launch the short ones behind it complete back to back and a waiter that is a
    // few microseconds late finds its slot reused.  ADVICE r5.)
    if ((int64_t)launch + kPartyEvents <= hi) return LGC_OK;
    // (no hipSetDevice: called from the notifier thread of the host while the main thread enqueues; events carry their device)
    RCHK(hipEventSynchronize(p->ev_done[launch % kPartyEvents]));      // kernel end = release: the tables are visible to the peer process
    return LGC_OK;
}
// ---- test hooks (tests/test_gpu_roles.py): what the peer-mapped ring holds at a given moment
// `launch` exactly as lgc_party_garble_ring issues it, in two halves: stage 1 = the record kernel (stops BEFORE the table
// pass of a critical-path launch), stage 2 = the table pass
extern "C" int lgc_test_party_garble_ring_stage(lgc_party *p, size_t launch, int stage, int *is_critical_path) {
    if (!p || !p->ring || p->role != LGC_ROLE_GARBLER || launch >= p->P.launches.size() || (stage != 1 && stage != 2))
        return lgc_fail(LGC_EINVAL, "bad argument");
    RCHK(hipSetDevice(p->device));
    bool crit = false;
    RCHK(party_launch<true>(p, p->P.launches[launch], ring_slot(p, launch), stage, &crit));
    RCHK(hipDeviceSynchronize());
    if (is_critical_path) *is_critical_path = crit ? 1 : 0;
    return LGC_OK;
}
extern "C" int lgc_test_party_ring_read(lgc_party *p, size_t launch, uint8_t *out, size_t bytes) {
    if (!p || !p->ring || !out || launch >= p->P.launches.size() ||
        bytes > (p->ring_bytes ? (size_t)p->P.launches[launch].steps * 2048 : p->ring_slot_bytes))
        return lgc_fail(LGC_EINVAL, "bad argument");
    RCHK(hipSetDevice(p->device));
    RCHK(hipMemcpy(out, ring_slot(p, launch), bytes, hipMemcpyDeviceToHost));
    return LGC_OK;
}
extern "C" int lgc_party_evaluate_ring(lgc_party *p, size_t launch) {
    if (!p) return lgc_fail(LGC_EINVAL, "null party");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "not the evaluator");
    if (!p->ring) return lgc_fail(LGC_ESTATE, "lgc_party_ring_open has not been called");
    if (!p->labels_ready) return lgc_fail(LGC_ESTATE, "input labels have not been set");
    if (launch >= p->P.launches.size()) return lgc_fail(LGC_EINVAL, "launch out of range");
    RCHK(hipSetDevice(p->device));
    RCHK(party_launch<false>(p, p->P.launches[launch], ring_slot(p, launch)));
    RCHK(hipDeviceSynchronize());
    return LGC_OK;
}

extern "C" int lgc_party_decode_bits(lgc_party *p, uint64_t *dec_out) {
    if (!p || !dec_out) return lgc_fail(LGC_EINVAL, "null argument");
    RCHK(hipSetDevice(p->device));
    RCHK(hipMemcpy(dec_out, p->dec, p->P.n_reveal * 8, hipMemcpyDeviceToHost));
    return LGC_OK;
}
extern "C" int lgc_party_finish(lgc_party *p, const uint64_t *garbler_dec, int64_t *beta, int64_t *trace, int64_t *inputs) {
    if (!p || !garbler_dec) return lgc_fail(LGC_EINVAL, "null argument");
    if (p->role != LGC_ROLE_EVALUATOR) return lgc_fail(LGC_ESTATE, "results are revealed to the evaluator (party 2) only");
    RCHK(hipSetDevice(p->device));
    RCHK(hipMemcpy(p->hdec.data(), p->dec, p->P.n_reveal * 8, hipMemcpyDeviceToHost));
    const Program &P = p->P;
    auto val = [&](uint32_t slot) -> int64_t {
        uint64_t v = p->hdec[slot] ^ garbler_dec[slot];
        return P.w == 32 ? (int64_t)(int32_t)(uint32_t)v : (int64_t)v;
    };
    if (beta)
        for (uint32_t t = 0; t < P.replicas; t++)
            for (size_t i = 0; i < P.d; i++) beta[(size_t)t * P.d + i] = val(P.rv_beta + t * P.reveal_stride + (uint32_t)i);
    if (trace && P.rv_trace != ~0u)
        for (size_t i = 0; i < (size_t)p->sys.num_iterations * (P.d + 4); i++) trace[i] = val(P.rv_trace + (uint32_t)i);
    if (inputs && P.rv_ab != ~0u)
        for (size_t i = 0; i < P.T + P.d; i++) inputs[i] = val(P.rv_ab + (uint32_t)i);
    return LGC_OK;
}

// ---------------------------------------------------------------- device buffers for host code (C)
// The host binaries are plain C: these give them device memory they can hand to the device-I/O forms of
// the OT calls (lgc_ot_*_set_device_io), and a hipIpc handle so that a peer process on the same node maps
// the buffer instead of receiving its bytes through a socket (u / y of the OT extension between two data
// providers: bin/linreg --ot_ring).  Allocations are zero-filled: a peer never sees stale HBM.
extern "C" int lgc_dev_alloc(int device, size_t bytes, void **ptr, uint8_t handle_out[64]) {
    if (!ptr || !bytes) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    void *p = 0;
    std::lock_guard<std::mutex> ipc_lock(ipc_mutex());
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return lgc_fail(LGC_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess && handle_out) {
        hipIpcMemHandle_t h;
        e = ipc_get_handle(&h, p);
        if (e == hipSuccess) memcpy(handle_out, &h, 64);
    }
    if (e != hipSuccess) { (void)hipFree(p); return lgc_fail(LGC_EHIP, "lgc_dev_alloc: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e)); }
    *ptr = p;
    return LGC_OK;
}
extern "C" void lgc_dev_free(void *ptr) { if (ptr) (void)hipFree(ptr); }
// for buffers that held secrets (label pairs): cleared before the memory goes back to the driver
extern "C" void lgc_dev_free_secret(void *ptr, size_t bytes) {
    if (!ptr) return;
    (void)hipMemset(ptr, 0, bytes);
    (void)hipDeviceSynchronize();
    (void)hipFree(ptr);
}
extern "C" int lgc_dev_open(int device, const uint8_t handle[64], void **ptr) {
    if (!handle || !ptr) return lgc_fail(LGC_EINVAL, "null argument");
    int rc = lgc_need_device(device);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void *p = 0;
    std::lock_guard<std::mutex> ipc_lock(ipc_mutex());
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return lgc_fail(LGC_EHIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *ptr = p;
    return LGC_OK;
}
extern "C" void lgc_dev_close(void *ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> ipc_lock(ipc_mutex());
    (void)hipIpcCloseMemHandle(ptr);
}
extern "C" int lgc_dev_upload(void *dst_dev, const void *src_host, size_t bytes) {
    if (!dst_dev || !src_host) return lgc_fail(LGC_EINVAL, "null argument");
    RCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return LGC_OK;
}
extern "C" int lgc_dev_download(void *dst_host, const void *src_dev, size_t bytes) {
    if (!dst_host || !src_dev) return lgc_fail(LGC_EINVAL, "null argument");
    RCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return LGC_OK;
}
