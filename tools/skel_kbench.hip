// tools/skel_kbench.hip — per-kernel latency of the skeleton chain kernels in a dependent launch chain
// (same stream, back to back), on a synthetic state: what one hist / k2 / rank launch costs including
// its launch gap.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pbwt_amd/csrc tools/skel_kbench.hip -o tools/skel_kbench
#include "pbwt_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace pbwtk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 0; }

template <int EPT>
static void run(int M, float p1, int reps) {
    const int T = 256 * EPT, W = (M + T - 1) / T, Wp = (W + 63) / 64 * 64, Mpad = (M + 4095) / 4096 * 4096;
    std::vector<int> a(Mpad), d(Mpad + 64);
    std::vector<unsigned char> keys(Mpad);
    std::vector<uint32_t> xT(Mpad);
    srand(7);
    for (int i = 0; i < Mpad; ++i) {
        a[i] = i < M ? (int)(((long long)i * 7919) % M) : 0; d[i] = rand() % 5000;
        unsigned k = 0, x = 0;
        for (int b = 0; b < 8; ++b) if ((rand() / (float)RAND_MAX) < p1) k |= 1u << b;
        for (int b = 0; b < 32; ++b) if ((rand() / (float)RAND_MAX) < p1) x |= 1u << b;
        keys[i] = (unsigned char)k; xT[i] = x;
    }
    int *dA, *dD, *dA2, *dD2, *tab; unsigned char *dK, *dK2; uint32_t *dX;
    CK(hipMalloc(&dA, Mpad * 4)); CK(hipMalloc(&dD, (Mpad + 64) * 4)); CK(hipMalloc(&dA2, Mpad * 4)); CK(hipMalloc(&dD2, (Mpad + 64) * 4));
    CK(hipMalloc(&dK, Mpad)); CK(hipMalloc(&dK2, Mpad)); CK(hipMalloc(&dX, Mpad * 4)); CK(hipMalloc(&tab, ((size_t)6 * (Wp + 64) * SKK + 1024) * 4));
    CK(hipMemcpy(dA, a.data(), Mpad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dD, d.data(), (Mpad + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dK, keys.data(), Mpad, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, xT.data(), Mpad * 4, hipMemcpyHostToDevice));
    SkArgs g;
    g.a = dA; g.d = dD; g.keys = dK; g.a_out = dA2; g.d_out = dD2; g.keys_out = dK2;
    g.tbl = (int2 *)tab; g.scan = (int2 *)tab + (size_t)(Wp + 8) * SKK; g.total = tab + (size_t)5 * (Wp + 8) * SKK;
    g.kbnext = (const unsigned char *)dX; g.has_next = 1; g.M = M; g.W = W; g.xcd = 0; g.pair = 0; g.tbl0 = nullptr; g.k = 100;
    Sk2Args k2; k2.tbl = g.tbl; k2.scan = g.scan; k2.total = g.total; k2.W = W;
    hipStream_t s;
    if (getenv("KB_PRIO")) { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, atoi(getenv("KB_PRIO")) ? hi : 0)); printf("stream: nonblocking, prio %s\n", getenv("KB_PRIO")); }
    else CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-28s %7.2f us/launch\n", name, 1e3 * ms / reps);
        return 1e3 * ms / reps;
    };
    printf("M=%d T=%d W=%d p1=%.2f\n", M, T, W, p1);
    timeit("empty", [&] { hipLaunchKernelGGL(empty_kernel, dim3(W), dim3(BLOCK), 0, s, (int *)nullptr); });
    float th = timeit("hist", [&] { hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, s, g); });
    auto k2l = [&] {
        if (W <= 256) hipLaunchKernelGGL((skel_k2_kernel<4, 4>), dim3(SKK / 4), dim3(BLOCK), 0, s, k2);   // one key per wave (16 / 8 keys per workgroup measured slower)
        else if (W <= 1024) hipLaunchKernelGGL((skel_k2_kernel<4, 16>), dim3(SKK / 4), dim3(BLOCK), 0, s, k2);
        else hipLaunchKernelGGL((skel_k2_kernel<2, 32>), dim3(SKK / 2), dim3(128), 0, s, k2);
    };
    float t2 = timeit("k2", k2l);
    float tr = timeit("rank", [&] { hipLaunchKernelGGL((skel_rank_kernel<EPT, 0>), dim3(W), dim3(BLOCK), 0, s, g); });
    float ta = timeit("hist+k2+rank", [&] { hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, s, g); k2l(); hipLaunchKernelGGL((skel_rank_kernel<EPT, 0>), dim3(W), dim3(BLOCK), 0, s, g); });
    printf("  sum %.2f, round %.2f us = %.2f us/site\n", th + t2 + tr, ta, ta / 8);
    {   // a real chain over NS ring slots (each round reads what the previous one scattered: nothing is L2-hot by accident)
        const int NS = 64;
        int *rA, *rD; unsigned char *rK;
        CK(hipMalloc(&rA, (size_t)(NS + 1) * Mpad * 4)); CK(hipMalloc(&rD, (size_t)(NS + 1) * (Mpad + 64) * 4)); CK(hipMalloc(&rK, (size_t)(NS + 1) * Mpad));
        CK(hipMemcpy(rA, a.data(), Mpad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rD, d.data(), (Mpad + 64) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rK, keys.data(), Mpad, hipMemcpyHostToDevice));
        int slot = 0;
        float tc = timeit("chain over 64 slots", [&] {
            SkArgs h = g;
            h.a = rA + (size_t)slot * Mpad; h.d = rD + (size_t)slot * (Mpad + 64); h.keys = rK + (size_t)slot * Mpad;
            h.a_out = rA + (size_t)(slot + 1) * Mpad; h.d_out = rD + (size_t)(slot + 1) * (Mpad + 64); h.keys_out = rK + (size_t)(slot + 1) * Mpad;
            h.kbnext = (const unsigned char *)dX + (size_t)(slot % 4) * 4096;
            hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, s, h); k2l(); hipLaunchKernelGGL((skel_rank_kernel<EPT, 0>), dim3(W), dim3(BLOCK), 0, s, h);
            slot = (slot + 1) % NS;
            if (slot == 0) { CK(hipMemcpyAsync(rA, rA + (size_t)NS * Mpad, Mpad * 4, hipMemcpyDeviceToDevice, s)); CK(hipMemcpyAsync(rD, rD + (size_t)NS * (Mpad + 64), (Mpad + 64) * 4, hipMemcpyDeviceToDevice, s)); CK(hipMemcpyAsync(rK, rK + (size_t)NS * Mpad, Mpad, hipMemcpyDeviceToDevice, s)); }
        });
        printf("  chain round %.2f us = %.2f us/site\n", tc, tc / 8);
        // the same 64 rounds captured once into a hipGraph and replayed
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int sl = 0; sl < NS; ++sl) {
            SkArgs h = g;
            h.a = rA + (size_t)sl * Mpad; h.d = rD + (size_t)sl * (Mpad + 64); h.keys = rK + (size_t)sl * Mpad;
            h.a_out = rA + (size_t)(sl + 1) * Mpad; h.d_out = rD + (size_t)(sl + 1) * (Mpad + 64); h.keys_out = rK + (size_t)(sl + 1) * Mpad;
            h.kbnext = (const unsigned char *)dX + (size_t)(sl % 4) * 4096;
            hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, s, h); k2l(); hipLaunchKernelGGL((skel_rank_kernel<EPT, 0>), dim3(W), dim3(BLOCK), 0, s, h);
        }
        CK(hipMemcpyAsync(rA, rA + (size_t)NS * Mpad, Mpad * 4, hipMemcpyDeviceToDevice, s)); CK(hipMemcpyAsync(rD, rD + (size_t)NS * (Mpad + 64), (Mpad + 64) * 4, hipMemcpyDeviceToDevice, s)); CK(hipMemcpyAsync(rK, rK + (size_t)NS * Mpad, Mpad, hipMemcpyDeviceToDevice, s));
        CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  graph replay of 64 rounds: round %.2f us = %.2f us/site\n", 1e3 * ms / (30 * NS), 1e3 * ms / (30 * NS) / 8);
    }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 100000, ept = argc > 2 ? atoi(argv[2]) : 2;
    const float p1 = argc > 3 ? atof(argv[3]) : 0.1f;
    const int reps = 2000;
    if (ept == 1) run<1>(M, p1, reps); else if (ept == 2) run<2>(M, p1, reps); else run<4>(M, p1, reps);
    return 0;
}
