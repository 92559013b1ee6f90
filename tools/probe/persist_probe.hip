// GPU box experiment: do two persistent kernels on two non-blocking streams run concurrently and see host-memory writes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
typedef unsigned long long u64;
struct Host { volatile u64 beatA, beatB, quit, flag, echo, rt0, rt1; };
__global__ void kA(Host *h) {
    u64 n = 0;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        n++;
        if (threadIdx.x == 0) __hip_atomic_store((u64 *)&h->beatA, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const u64 f = __hip_atomic_load((u64 *)&h->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) __hip_atomic_store((u64 *)&h->echo, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__hip_atomic_load((u64 *)&h->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) { h->rt0 = t0; h->rt1 = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void kB(Host *h) {
    u64 n = 0;
    for (;;) {
        n++;
        if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((u64 *)&h->beatB, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__hip_atomic_load((u64 *)&h->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main() {
    Host *h; hipHostMalloc((void **)&h, sizeof(Host), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    memset((void *)h, 0, sizeof(Host));
    Host *hd; hipHostGetDevicePointer((void **)&hd, h, 0);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    printf("host %p dev %p\n", (void *)h, (void *)hd);
    hipLaunchKernelGGL(kA, dim3(1), dim3(64), 0, sa, hd);
    printf("launched A: %s\n", hipGetErrorString(hipGetLastError()));
    usleep(100000);
    printf("after 100 ms: beatA %llu\n", h->beatA);
    hipLaunchKernelGGL(kB, dim3(256), dim3(576), 81776, sb, hd);
    printf("launched B: %s\n", hipGetErrorString(hipGetLastError()));
    usleep(100000);
    printf("after 200 ms: beatA %llu beatB %llu\n", h->beatA, h->beatB);
    // flag round trip
    double best = 1e9, sum = 0;
    for (int i = 1; i <= 200; i++) {
        const double t0 = now();
        h->flag = i;
        while (h->echo != (u64)i) {}
        const double dt = now() - t0;
        sum += dt; if (dt < best) best = dt;
    }
    printf("host->device->host flag round trip: best %.2f us, mean %.2f us\n", best * 1e6, sum / 200 * 1e6);
    const double tq = now();
    h->quit = 1;
    hipStreamSynchronize(sa); hipStreamSynchronize(sb);
    printf("quit -> both streams idle: %.1f us; memrealtime ticks per second: %.0f\n", (now() - tq) * 1e6, (double)(h->rt1 - h->rt0) / 0.2);
    return 0;
}
