// latprobe2.hip — why does a dependent 800 KB copy kernel cost 20 us?  separate read / write / reuse effects
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const int4 *in, int *sink, int n4) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { int4 v = in[i]; if (v.x == 0x7fffffff) *sink = 1; }
}
__global__ __launch_bounds__(256) void k_write(int4 *out, int n4, int val) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = make_int4(val, i, 0, 0);
}
__global__ __launch_bounds__(256) void k_copy(const int4 *in, int4 *out, int n4) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { int4 v = in[i]; v.x += 1; out[i] = v; }
}
// grid-stride copy with few WGs
__global__ __launch_bounds__(256) void k_copy_gs(const int4 *in, int4 *out, int n4) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) { int4 v = in[i]; v.x += 1; out[i] = v; }
}
// scatter copy: out[perm(i)] = in[i] (4-byte elements, like the partition scatter)
__global__ __launch_bounds__(256) void k_scatter(const int *in, int *out, int n, int half) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { int v = in[i]; int dst = (i & 1) ? half + (i >> 1) : (i >> 1); out[dst] = v + 1; }
}

template <typename F>
static float time_graph(hipStream_t st, int reps, int nodes, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch(i);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
    (void)hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(b, st); (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms * 1e3f / (reps * nodes);
}

int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NODES = 256, REPS = 8;
    int *sink; CK(hipMalloc(&sink, 4));
    for (int kb : {64, 800, 8000, 64000}) {
        const int n4 = kb * 1024 / 16; const int nb = (n4 + 255) / 256;
        int4 *b0, *b1; CK(hipMalloc(&b0, (size_t)n4 * 16)); CK(hipMalloc(&b1, (size_t)n4 * 16));
        CK(hipMemset(b0, 0, (size_t)n4 * 16)); CK(hipMemset(b1, 0, (size_t)n4 * 16));
        printf("---- %d KB (%d WGs)\n", kb, nb);
        printf("read-only same buffer        : %.2f us\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, st, b0, sink, n4); }));
        printf("write-only same buffer       : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, st, b0, n4, i); }));
        printf("write b0 then read b0 (pairs): %.2f us per kernel\n", time_graph(st, REPS, NODES, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, st, b0, sink, n4); else hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, st, b0, n4, i); }));
        printf("copy b0->b1 fixed            : %.2f us\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_copy, dim3(nb), dim3(256), 0, st, b0, b1, n4); }));
        printf("copy ping-pong               : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy, dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n4); }));
        printf("copy ping-pong grid-stride 32 WG : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy_gs, dim3(32), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n4); }));
        printf("copy ping-pong grid-stride 8 WG  : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy_gs, dim3(8), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n4); }));
        printf("scatter ping-pong (4B)       : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_scatter, dim3((n4 * 4 + 255) / 256), dim3(256), 0, st, (const int *)((i & 1) ? b1 : b0), (int *)((i & 1) ? b0 : b1), n4 * 4, n4 * 2); }));
        (void)hipFree(b0); (void)hipFree(b1);
    }
    return 0;
}
