// latprobe3.hip — isolate why load+store kernels are slow across a dependent launch chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy16(const int4 *in, int4 *out, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { int4 v = in[i]; v.x += 1; out[i] = v; } }
__global__ __launch_bounds__(256) void k_copy8(const int2 *in, int2 *out, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { int2 v = in[i]; v.x += 1; out[i] = v; } }
__global__ __launch_bounds__(256) void k_copy4(const int *in, int *out, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { out[i] = in[i] + 1; } }
__global__ __launch_bounds__(256) void k_copy16_nt(const int4 *in, int4 *out, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { int4 v = in[i]; v.x += 1; __builtin_nontemporal_store(v.x, &out[i].x); __builtin_nontemporal_store(v.y, &out[i].y); __builtin_nontemporal_store(v.z, &out[i].z); __builtin_nontemporal_store(v.w, &out[i].w); } }
// half the WGs read, the other half write: no load->store dependency
__global__ __launch_bounds__(256) void k_split(const int4 *in, int4 *out, int *sink, int n) {
    int b = blockIdx.x >> 1; int i = b * 256 + threadIdx.x;
    if (i < n) { if (blockIdx.x & 1) { int4 v = in[i]; if (v.x == 0x7fffffff) *sink = 1; } else out[i] = make_int4(i, 1, 2, 3); }
}
// store does not depend on the load but both are in the same thread
__global__ __launch_bounds__(256) void k_indep(const int4 *in, int4 *out, int *sink, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { int4 v = in[i]; out[i] = make_int4(i, 1, 2, 3); if (v.x == 0x7fffffff) *sink = 1; }
}
// in-place update
__global__ __launch_bounds__(256) void k_inplace(int4 *io, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { int4 v = io[i]; v.x += 1; io[i] = v; } }

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
template <typename F>
static float time_eager(hipStream_t st, int n, F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 64; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(a, st);
    for (int i = 0; i < n; ++i) launch(i);
    (void)hipEventRecord(b, st); (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NODES = 256, REPS = 8;
    int *sink; CK(hipMalloc(&sink, 4));
    const size_t bytes = 800 * 1024;
    char *pool; CK(hipMalloc(&pool, 64 << 20)); CK(hipMemset(pool, 0, 64 << 20));
    int4 *b0 = (int4 *)pool;
    const int n16 = bytes / 16, nb16 = (n16 + 255) / 256;
    for (size_t off : {(size_t)bytes, (size_t)(1 << 20), (size_t)(1 << 20) + 4096 + 256, (size_t)(8 << 20), (size_t)(8 << 20) + 64 * 37}) {
        int4 *b1 = (int4 *)(pool + off);
        printf("copy16 fixed, out = in + %zu bytes : %.2f us\n", off, time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_copy16, dim3(nb16), dim3(256), 0, st, b0, b1, n16); }));
    }
    int4 *b1 = (int4 *)(pool + (8 << 20));
    printf("copy16 ping-pong graph : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy16, dim3(nb16), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n16); }));
    printf("copy16 ping-pong eager : %.2f us\n", time_eager(st, 4096, [&](int i) { hipLaunchKernelGGL(k_copy16, dim3(nb16), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n16); }));
    printf("copy8  ping-pong graph : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy8, dim3((n16 * 2 + 255) / 256), dim3(256), 0, st, (const int2 *)((i & 1) ? b1 : b0), (int2 *)((i & 1) ? b0 : b1), n16 * 2); }));
    printf("copy4  ping-pong graph : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy4, dim3((n16 * 4 + 255) / 256), dim3(256), 0, st, (const int *)((i & 1) ? b1 : b0), (int *)((i & 1) ? b0 : b1), n16 * 4); }));
    printf("copy16 nontemporal store ping-pong : %.2f us\n", time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy16_nt, dim3(nb16), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n16); }));
    printf("split readers/writers  : %.2f us\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_split, dim3(nb16 * 2), dim3(256), 0, st, b0, b1, sink, n16); }));
    printf("indep load+store same thread : %.2f us\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_indep, dim3(nb16), dim3(256), 0, st, b0, b1, sink, n16); }));
    printf("in-place update        : %.2f us\n", time_graph(st, REPS, NODES, [&](int) { hipLaunchKernelGGL(k_inplace, dim3(nb16), dim3(256), 0, st, b0, n16); }));
    // how does copy16 scale with bytes (WG count)?
    for (int kb : {16, 64, 128, 256, 400, 800, 1600}) {
        int n = kb * 1024 / 16, nb = (n + 255) / 256;
        printf("copy16 ping-pong %5d KB (%4d WGs): %.2f us   copy4: %.2f us\n", kb, nb,
               time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy16, dim3(nb), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, n); }),
               time_graph(st, REPS, NODES, [&](int i) { hipLaunchKernelGGL(k_copy4, dim3((n * 4 + 255) / 256), dim3(256), 0, st, (const int *)((i & 1) ? b1 : b0), (int *)((i & 1) ? b0 : b1), n * 4); }));
    }
    return 0;
}
