// clkprobe.hip — effective shader clock inside short kernels of a dependent launch chain vs a busy GPU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
// dependent integer chain: n iterations of 8 dependent v_add/xor (approx 8 * 4..8 cycles each)
__global__ void k_alu(unsigned long long *out, int n, int slot) {
    unsigned x = threadIdx.x + 1;
    unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        x = x * 3 + 1; x ^= x >> 3; x = x * 5 + 7; x ^= x >> 5; x = x * 3 + 1; x ^= x >> 3; x = x * 5 + 7; x ^= x >> 5;
    }
    unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[slot * 2] = w1 - w0; out[slot * 2 + 1] = x; }
}
__global__ void k_heat(float *p, int iters) {   // keep many CUs busy
    float a = threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 1.00001f + a * 1e-9f; }
    if (a == 12345.f) p[0] = a + b;
}
__global__ void k_empty() {}
int main() {
    hipStream_t st, st2; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    unsigned long long *d; CK(hipMalloc(&d, 4096 * 16)); float *hp; CK(hipMalloc(&hp, 64));
    std::vector<unsigned long long> h(4096 * 2);
    const int n = 2000;
    auto report = [&](const char *name, int cnt) {
        CK(hipMemcpy(h.data(), d, cnt * 16, hipMemcpyDeviceToHost));
        double mn = 1e30, mx = 0, av = 0; for (int i = 0; i < cnt; ++i) { double v = h[i * 2] * 10.0; mn = v < mn ? v : mn; mx = v > mx ? v : mx; av += v; }
        printf("%-52s: ALU loop (n=%d) min %.0f avg %.0f max %.0f ns\n", name, n, mn, av / cnt, mx); return 0; };
    // (a) isolated single kernels from idle
    for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(k_alu, dim3(1), dim3(64), 0, st, d, n, i); CK(hipStreamSynchronize(st)); }
    report("isolated launches from idle (1 wave)", 8);
    // (b) inside a dense chain of small kernels (98 WGs each)
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_alu, dim3(98), dim3(256), 0, st, d, n, i);
    CK(hipStreamSynchronize(st)); report("chain of 2000 launches x 98 WGs", 2000);
    // (c) chain while a heater keeps the rest of the chip busy on another stream
    hipLaunchKernelGGL(k_heat, dim3(2048), dim3(256), 0, st2, hp, 4000000);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_alu, dim3(98), dim3(256), 0, st, d, n, i);
    CK(hipStreamSynchronize(st)); report("same chain beside a 2048-WG heater kernel", 2000);
    CK(hipStreamSynchronize(st2));
    // (d) big kernel: all CUs busy with the ALU loop
    hipLaunchKernelGGL(k_alu, dim3(4096), dim3(256), 0, st, d, n * 50, 0); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost));
    printf("%-52s: ALU loop (n=%d) %.0f ns => per n=%d: %.0f ns\n", "one 4096-WG kernel (chip busy)", n * 50, h[0] * 10.0, n, h[0] * 10.0 / 50);
    // (e) chain again right after the busy kernel
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_alu, dim3(98), dim3(256), 0, st, d, n, i);
    CK(hipStreamSynchronize(st)); report("chain right after the busy kernel", 2000);
    CK(hipMemcpy(h.data(), d, 2000 * 16, hipMemcpyDeviceToHost));
    printf("   first 5: %.0f %.0f %.0f %.0f %.0f  last: %.0f ns\n", h[0] * 10.0, h[2] * 10.0, h[4] * 10.0, h[6] * 10.0, h[8] * 10.0, h[3998] * 10.0);
    return 0;
}
