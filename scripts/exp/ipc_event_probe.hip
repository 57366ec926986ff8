// ipc_event_probe -- do interprocess HIP events order work across two processes on one GPU?  (A device-side hand-off between
// the CSP and the Evaluator process would need them: hipStreamWaitEvent on the other process's event instead of a
// hipDeviceSynchronize + socket token per launch.)  Parent = producer, child = consumer; fork happens BEFORE any HIP call.
// The producer runs a slow kernel that fills a buffer shared through hipIpc and records event k; the consumer waits for
// event k on its stream and checks the buffer with a kernel.  Rounds reuse a pool of events.  Run under `timeout 60`.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <time.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d, pid %d)\n", #x, hipGetErrorString(e_), __LINE__, getpid()); exit(2); } } while (0)
static void rd(int fd, void *p, size_t n) { char *c = (char *)p; while (n) { ssize_t k = read(fd, c, n); if (k <= 0) { fprintf(stderr, "pipe closed\n"); exit(3); } c += k; n -= (size_t)k; } }
static void wr(int fd, const void *p, size_t n) { if (write(fd, p, n) != (ssize_t)n) { fprintf(stderr, "pipe write\n"); exit(3); } }
__global__ void fill(unsigned *buf, unsigned n, unsigned val, int spin) {
    // slow on purpose: the consumer must really wait
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = val;
}
__global__ void check(const unsigned *buf, unsigned n, unsigned val, unsigned *bad) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (buf[i] != val) atomicAdd(bad, 1u);
}
int main() {
    const int K = 8, ROUNDS = 64; const unsigned N = 1u << 22;
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 1;
    pid_t pid = fork();
    if (pid == 0) {                                   // consumer
        hipStream_t st; CK(hipSetDevice(0)); CK(hipStreamCreate(&st));
        hipIpcMemHandle_t mh; rd(ab[0], &mh, sizeof mh);
        unsigned *buf; CK(hipIpcOpenMemHandle((void **)&buf, mh, hipIpcMemLazyEnablePeerAccess));
        hipIpcEventHandle_t eh[K]; hipEvent_t ev[K];
        struct timespec ta, tb; clock_gettime(CLOCK_MONOTONIC, &ta);
        for (int k = 0; k < K; k++) { rd(ab[0], &eh[k], sizeof eh[k]); CK(hipIpcOpenEventHandle(&ev[k], eh[k])); }
        clock_gettime(CLOCK_MONOTONIC, &tb);
        printf("consumer: %d event handles received + opened in %.2f ms\n", K, ((tb.tv_sec - ta.tv_sec) * 1e3 + (tb.tv_nsec - ta.tv_nsec) * 1e-6));
        unsigned *bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        unsigned total_bad = 0;
        for (int r = 0; r < ROUNDS; r++) {
            char go; rd(ab[0], &go, 1);               // the producer has ENQUEUED round r (record issued, kernel maybe still running)
            CK(hipStreamWaitEvent(st, ev[r % K], 0));
            hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, st, buf, N, (unsigned)(r + 1), bad);
            CK(hipStreamSynchronize(st));
            unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemset(bad, 0, 4));
            total_bad += h ? 1 : 0;
            wr(ba[1], &go, 1);                        // round r consumed: its buffer may be overwritten
        }
        printf("consumer: %d rounds, %u rounds saw stale data\n", ROUNDS, total_bad);
        return total_bad ? 4 : 0;
    }
    hipStream_t st; CK(hipSetDevice(0)); CK(hipStreamCreate(&st));
    unsigned *buf; CK(hipMalloc(&buf, N * 4)); CK(hipMemset(buf, 0, N * 4));
    hipIpcMemHandle_t mh; CK(hipIpcGetMemHandle(&mh, buf)); wr(ab[1], &mh, sizeof mh);
    hipEvent_t ev[K];
    for (int k = 0; k < K; k++) {
        CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming | hipEventInterprocess));
        hipIpcEventHandle_t eh; CK(hipIpcGetEventHandle(&eh, ev[k])); wr(ab[1], &eh, sizeof eh);
    }
    for (int r = 0; r < ROUNDS; r++) {
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, buf, N, (unsigned)(r + 1), 2000000);   // ~1 ms of spinning first
        CK(hipEventRecord(ev[r % K], st));
        char go = 1; wr(ab[1], &go, 1);               // no host synchronisation before the consumer is told
        rd(ba[0], &go, 1);
    }
    CK(hipDeviceSynchronize());
    {   // what a pool of events costs: create + export (the consumer's open is timed in the first loop above per event too)
        struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
        const int M = 1000; static hipEvent_t pool[1000];
        for (int k = 0; k < M; k++) { CK(hipEventCreateWithFlags(&pool[k], hipEventDisableTiming | hipEventInterprocess)); hipIpcEventHandle_t eh; CK(hipIpcGetEventHandle(&eh, pool[k])); }
        clock_gettime(CLOCK_MONOTONIC, &b);
        printf("producer: %d interprocess events created + exported in %.2f ms\n", M, ((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6));
        clock_gettime(CLOCK_MONOTONIC, &a);
        for (int k = 0; k < M; k++) CK(hipEventRecord(pool[k], st));
        CK(hipStreamSynchronize(st));
        clock_gettime(CLOCK_MONOTONIC, &b);
        printf("producer: %d records + sync in %.2f ms\n", M, ((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6));
        clock_gettime(CLOCK_MONOTONIC, &a);
        for (int k = 0; k < M; k++) CK(hipEventDestroy(pool[k]));
        clock_gettime(CLOCK_MONOTONIC, &b);
        printf("producer: %d destroyed in %.2f ms\n", M, ((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6));
    }
    int stt = 0; waitpid(pid, &stt, 0);
    printf("producer done; consumer exit %d\n", WIFEXITED(stt) ? WEXITSTATUS(stt) : -1);
    return WIFEXITED(stt) ? WEXITSTATUS(stt) : 5;
}
