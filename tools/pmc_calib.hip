// pmc_calib.hip — known-byte-count kernels in the step kernel's access pattern (4 B per lane,
// coalesced) to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void calib_copy4(const int *in, int *out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;
}
__global__ __launch_bounds__(256) void calib_read4(const int *in, int *sink, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && in[i] == 0x7fffffff) *sink = 1;
}
int main() {
    const size_t n = (size_t)256 << 20;           // 1 GiB per buffer: larger than the 256 MiB Infinity Cache
    int *a, *b, *s;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess || hipMalloc(&s, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(calib_copy4, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(calib_read4, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, s, n);
    }
    hipDeviceSynchronize();
    printf("calib: copy4 reads %zu bytes writes %zu bytes; read4 reads %zu bytes\n", n * 4, n * 4, n * 4);
    return 0;
}
