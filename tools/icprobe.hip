// icprobe.hip — is the instruction cache cold at every launch?  straight-line code vs a loop doing the same ALU work
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
template <int N> __global__ void k_straight(unsigned *out, unsigned seed) {
    unsigned x = threadIdx.x + seed;
#pragma unroll
    for (int i = 0; i < N; ++i) { x = x * 3u + (unsigned)i; x ^= x >> 3; }
    if (x == 0x12345u) out[0] = x;
}
__global__ void k_loop(unsigned *out, unsigned seed, int n) {
    unsigned x = threadIdx.x + seed;
#pragma unroll 1
    for (int i = 0; i < n; ++i) { x = x * 3u + (unsigned)i; x ^= x >> 3; }
    if (x == 0x12345u) out[0] = x;
}
template <typename F> static float chain(hipStream_t st, int n, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) f(i);
    hipStreamSynchronize(st); hipEventRecord(a, st);
    for (int i = 0; i < n; ++i) f(i);
    hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / n;
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *d; CK(hipMalloc(&d, 64));
    const int G = 98;
    printf("loop     n=256 : %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL(k_loop, dim3(G), dim3(256), 0, st, d, i, 256); }));
    printf("straight N=256 : %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL((k_straight<256>), dim3(G), dim3(256), 0, st, d, i); }));
    printf("loop     n=1024: %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL(k_loop, dim3(G), dim3(256), 0, st, d, i, 1024); }));
    printf("straight N=1024: %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL((k_straight<1024>), dim3(G), dim3(256), 0, st, d, i); }));
    printf("loop     n=4096: %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL(k_loop, dim3(G), dim3(256), 0, st, d, i, 4096); }));
    printf("straight N=4096: %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL((k_straight<4096>), dim3(G), dim3(256), 0, st, d, i); }));
    // alternate two different straight kernels (evicts nothing if the cache is big enough)
    printf("straight N=1024 alternating with N=256: %.2f us/launch\n", chain(st, 2000, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_straight<1024>), dim3(G), dim3(256), 0, st, d, i); else hipLaunchKernelGGL((k_straight<256>), dim3(G), dim3(256), 0, st, d, i); }));
    printf("straight N=1024, 1 WG : %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL((k_straight<1024>), dim3(1), dim3(64), 0, st, d, i); }));
    printf("straight N=1024, 1024 WG : %.2f us/launch\n", chain(st, 2000, [&](int i) { hipLaunchKernelGGL((k_straight<1024>), dim3(1024), dim3(256), 0, st, d, i); }));
    return 0;
}
