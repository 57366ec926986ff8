// hip_init_probe -- what the first HIP calls of a process cost (start-up of the end-to-end runs).
// Build: hipcc --offload-arch=gfx950 -O2 -o bin/hip_init_probe hip_init_probe.hip ; run several at once to see the contention.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv) {
    double t0 = now(), t;
    int n = 0;
    hipGetDeviceCount(&n);              t = now(); printf("hipGetDeviceCount %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipSetDevice(0);                    t = now(); printf("hipSetDevice      %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipFree(0);                         t = now(); printf("hipFree(0)        %.1f ms\n", (t - t0) * 1e3); t0 = t;
    int *d = 0; hipMalloc(&d, 1 << 20); t = now(); printf("hipMalloc 1 MiB   %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipMemset(d, 0, 1 << 20); hipDeviceSynchronize(); t = now(); printf("hipMemset + sync  %.1f ms\n", (t - t0) * 1e3); t0 = t;
    static int h[256]; hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); t = now(); printf("hipMemcpy H2D     %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); t = now(); printf("first kernel      %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipStream_t s; hipStreamCreate(&s); t = now(); printf("hipStreamCreate   %.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); t = now(); printf("kernel on stream  %.1f ms\n", (t - t0) * 1e3); t0 = t;
    size_t big = argc > 1 ? (size_t)atof(argv[1]) : 0;
    if (big) {
        void *r = 0; hipMalloc(&r, big); t = now(); printf("hipMalloc %zu MB  %.1f ms\n", big >> 20, (t - t0) * 1e3); t0 = t;
        hipMemset(r, 0, big); hipDeviceSynchronize(); t = now(); printf("memset of it      %.1f ms\n", (t - t0) * 1e3); t0 = t;
        hipFree(r); t = now(); printf("hipFree of it     %.1f ms\n", (t - t0) * 1e3); t0 = t;
    }
    return 0;
}
