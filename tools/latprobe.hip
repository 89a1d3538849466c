// latprobe.hip — micro-measurements that decide the step-kernel structure:
// dependent-kernel cost, dependent-load cost inside a kernel, shader clock under a launch chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_clock(unsigned long long *out) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    // spin ~20 us of wall clock
    while (wall_clock64() - w0 < 2000) {}
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
// n dependent loads (pointer chase), one lane
__global__ void k_chase(const int *next, int start, int n, int *sink) {
    int p = start;
    for (int i = 0; i < n; ++i) p = next[p];
    if (p == -12345) *sink = p;
}
// ping-pong tile copy with `hops` extra dependent loads through a small table
template <int HOPS>
__global__ __launch_bounds__(256) void k_copy(const int4 *in, int4 *out, const int *table, int n4) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int off = 0;
#pragma unroll
    for (int h = 0; h < HOPS; ++h) off = table[(off + threadIdx.x + h * 64) & 1023];   // dependent chain, values are 0
    if (i < n4) { int4 v = in[i + off]; v.x += 1; out[i] = v; }
}
__global__ __launch_bounds__(256) void k_copy_atomic(const int4 *in, int4 *out, int *acc, int n4) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { int4 v = in[i]; v.x += 1; out[i] = v; }
    if (threadIdx.x < 16) atomicAdd(acc + ((blockIdx.x + threadIdx.x) & 127), 1);
}

template <typename F>
static float time_graph(hipStream_t st, int reps, int nodes, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * nodes);       // us per node
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NODES = 512, REPS = 8;
    printf("empty kernel chain           : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_empty, dim3(98), dim3(256), 0, st); }));
    printf("empty kernel chain (1 WG)    : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
    printf("empty kernel chain (1024 WG) : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st); }));
    // shader clock while a chain is running and idle
    unsigned long long *dclk; CK(hipMalloc(&dclk, 16));
    unsigned long long hclk[2];
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, dclk); CK(hipMemcpyAsync(hclk, dclk, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("clock64/wall_clock64 ratio (cold): %.3f  (wall=100MHz => shader clock ~ %.0f MHz if clock64 counts shader cycles)\n", (double)hclk[0] / hclk[1], 100.0 * hclk[0] / hclk[1]);
    for (int r = 0; r < 3; ++r) { for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k_empty, dim3(98), dim3(256), 0, st);
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, dclk); CK(hipMemcpyAsync(hclk, dclk, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        printf("clock ratio after 20000-launch chain: %.3f\n", (double)hclk[0] / hclk[1]); }
    // pointer chase: per-hop latency, cold lines (stride 4 KB over 64 MB)
    const int NP = 1 << 24;
    std::vector<int> h(NP, 0);
    for (int i = 0; i < NP; ++i) h[i] = (int)(((long long)i + 1024 * 37 + 1) % NP);
    int *dnext, *dsink; CK(hipMalloc(&dnext, NP * 4)); CK(hipMalloc(&dsink, 4));
    CK(hipMemcpy(dnext, h.data(), NP * 4, hipMemcpyHostToDevice));
    for (int n : {0, 1, 2, 4, 8, 16, 32}) {
        float us = time_graph(st, 4, 64, [&](int i) { hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, st, dnext, (i * 7919 * 1024) % NP, n, dsink); });
        printf("chase n=%2d dependent loads (far lines): %.2f us/kernel\n", n, us);
    }
    // same lines every kernel (L2-resident)
    for (int n : {1, 8, 32}) {
        float us = time_graph(st, 4, 64, [&](int) { hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, st, dnext, 0, n, dsink); });
        printf("chase n=%2d dependent loads (same lines each kernel): %.2f us/kernel\n", n, us);
    }
    // ping-pong copy of 800 KB (100k ints x 2 arrays) across 98 WGs
    const int n4 = 50000;   // int4 = 800 KB
    int4 *b0, *b1; int *table, *acc;
    CK(hipMalloc(&b0, n4 * 16 + 65536)); CK(hipMalloc(&b1, n4 * 16 + 65536)); CK(hipMalloc(&table, 4096)); CK(hipMalloc(&acc, 4096));
    CK(hipMemset(b0, 0, n4 * 16 + 65536)); CK(hipMemset(b1, 0, n4 * 16 + 65536)); CK(hipMemset(table, 0, 4096)); CK(hipMemset(acc, 0, 4096));
    const int nb = (n4 + 255) / 256;
    printf("copy 800KB ping-pong, 0 extra hops : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL((k_copy<0>), dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, table, n4); }));
    printf("copy 800KB ping-pong, 1 extra hop  : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL((k_copy<1>), dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, table, n4); }));
    printf("copy 800KB ping-pong, 2 extra hops : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL((k_copy<2>), dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, table, n4); }));
    printf("copy 800KB ping-pong, 4 extra hops : %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL((k_copy<4>), dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, table, n4); }));
    printf("copy 800KB ping-pong + 16 atomics/WG: %.2f us/kernel\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy_atomic, dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, acc, n4); }));
    // eager (no graph) launch rate
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st);
      for (int i = 0; i < 4096; ++i) hipLaunchKernelGGL((k_copy<0>), dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, table, n4);
      hipEventRecord(b, st); hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, a, b);
      printf("copy 800KB ping-pong eager (no graph): %.2f us/kernel\n", ms * 1e3f / 4096); }
    return 0;
}
