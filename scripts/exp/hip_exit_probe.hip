// hip_exit_probe -- what a process pays AFTER main: HIP runtime exit handlers / kernel-driver cleanup, as a function of what it
// created.  argv: <n_streams> <alloc_mb> <free_alloc 0|1> <exit_kind 0 return | 1 _exit | 2 hipDeviceReset then return>.
// Prints CLOCK_MONOTONIC at the end of main; the parent (hip_exit_probe.py) measures when the process is gone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv) {
    int ns = argc > 1 ? atoi(argv[1]) : 0, mb = argc > 2 ? atoi(argv[2]) : 0, fr = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 4 ? atoi(argv[4]) : 0;
    int *d = 0;
    hipSetDevice(0);
    hipMalloc(&d, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    hipStream_t st[64];
    for (int i = 0; i < ns && i < 64; i++) { hipStreamCreate(&st[i]); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st[i], d); }
    hipDeviceSynchronize();
    void *big = 0;
    if (mb) { hipMalloc(&big, (size_t)mb << 20); hipMemset(big, 0, (size_t)mb << 20); hipDeviceSynchronize(); if (fr) hipFree(big); }
    if (kind == 2) hipDeviceReset();
    printf("%.6f\n", now());
    fflush(stdout);
    if (kind == 1) _exit(0);
    return 0;
}
