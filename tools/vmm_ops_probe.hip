#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void fill_kernel(unsigned char *p, size_t n, int rep) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)((i * 2654435761u + rep) >> 7); }
__global__ void sum_kernel(const unsigned char *p, size_t n, unsigned long long *out) { unsigned long long s = 0; for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i] * (i % 251 + 1); atomicAdd(out, s); }
static hipMemAllocationProp prop; static size_t gran;
static int vmm_alloc(size_t n, unsigned char **out) {
    void *va; hipMemGenericAllocationHandle_t h; size_t mapped = (n + gran - 1) / gran * gran;
    CK(hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0)); CK(hipMemCreate(&h, mapped, &prop, 0)); CK(hipMemMap((char *)va + gran, mapped, 0, h, 0));
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess((char *)va + gran, mapped, &ad, 1));
    *out = (unsigned char *)va + gran + (mapped - (n + 255) / 256 * 256); return 0;
}
int main() {
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *dsum; CK(hipMalloc((void **)&dsum, 8));
    int bad[4] = {0, 0, 0, 0};
    for (size_t n : {size_t(1000), size_t(70001), size_t(3 << 20) + 17, size_t(40 << 20) + 5}) for (int rep = 0; rep < 20; ++rep) {
        unsigned char *a, *b; if (vmm_alloc(n, &a) || vmm_alloc(n, &b)) return 2;
        std::vector<unsigned char> host(n), back(n); unsigned long long want = 0, want0 = 0;
        for (size_t i = 0; i < n; ++i) { host[i] = (unsigned char)((i * 2654435761u + rep) >> 7); want += host[i] * (i % 251 + 1); want0 += 0x5a * (i % 251 + 1); }
        unsigned long long got = 0;
        // memset
        CK(hipMemsetAsync(dsum, 0, 8, st)); CK(hipMemsetAsync(a, 0x5a, n, st)); hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, a, n, dsum);
        CK(hipMemcpyAsync(&got, dsum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); if (got != want0) ++bad[0];
        // D2D (kernel-filled source)
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, st, a, n, rep);
        CK(hipMemsetAsync(dsum, 0, 8, st)); CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, st)); hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, b, n, dsum);
        CK(hipMemcpyAsync(&got, dsum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); if (got != want) ++bad[1];
        // D2H to pageable
        CK(hipMemcpyAsync(back.data(), a, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); if (back != host) ++bad[2];
        // synchronous hipMemcpy H2D
        CK(hipMemcpy(b, host.data(), n, hipMemcpyHostToDevice)); CK(hipMemsetAsync(dsum, 0, 8, st)); hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, b, n, dsum);
        CK(hipMemcpyAsync(&got, dsum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); if (got != want) ++bad[3];
    }
    printf("VMM memory, 80 ops each: memset bad %d, D2D bad %d, D2H bad %d, sync H2D bad %d\n", bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
