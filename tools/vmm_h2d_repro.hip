// vmm_h2d_repro.hip — does a host-to-device copy INTO memory mapped through the virtual-memory API (hipMemCreate + hipMemMap: what
// PBWTAMD_GUARD=1 allocates, so that an unmapped page sits right behind every buffer) deliver the bytes?  No pbwt code: one buffer, one
// copy from pageable host memory, one checksum kernel — into plain hipMalloc memory (pass 0), into VMM memory stream-ordered (1), into
// VMM memory with the host waiting for the copy before it enqueues the kernel (2), and into VMM memory through a hipMalloc bounce buffer
// and a device copy kernel (3).  Measured on this image (ROCm 7.2 box, MI355X): passes 1 AND 2 lose half of the 3 MB copies (every other
// repetition: the bytes never arrive, waiting does not help), passes 0 and 3 never do.  Exit status = mismatches of pass 3.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/vmm_h2d_repro tools/vmm_h2d_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void copy_kernel(unsigned char *dst, const unsigned char *src, size_t n) {
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void sum_kernel(const unsigned char *p, size_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i] * (i % 251 + 1);
    atomicAdd(out, s);
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *dsum; CK(hipMalloc((void **)&dsum, 8));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    int badv[4] = {0, 0, 0, 0};
    for (int vmm = 0; vmm < 4; ++vmm) for (size_t n : {size_t(1000), size_t(70001), size_t(3 << 20) + 17, size_t(40 << 20) + 5}) for (int rep = 0; rep < 20; ++rep) {
        unsigned char *dst = nullptr; void *va = nullptr; hipMemGenericAllocationHandle_t h; size_t mapped = (n + gran - 1) / gran * gran;
        const bool sync_after_copy = vmm == 2;                 // third pass: VMM memory again, the host waits for the copy before it enqueues the kernel
        if (!vmm) CK(hipMalloc((void **)&dst, n));
        else {
            CK(hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0)); CK(hipMemCreate(&h, mapped, &prop, 0));
            CK(hipMemMap((char *)va + gran, mapped, 0, h, 0));
            hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess((char *)va + gran, mapped, &ad, 1));
            dst = (unsigned char *)va + gran + (mapped - (n + 255) / 256 * 256);          // ends (to 256 bytes) at the end of the mapping: PBWTAMD_GUARD=1's placement
        }
        std::vector<unsigned char> host(n); unsigned long long want = 0;
        for (size_t i = 0; i < n; ++i) { host[i] = (unsigned char)((i * 2654435761u + rep) >> 7); want += host[i] * (i % 251 + 1); }
        CK(hipMemsetAsync(dsum, 0, 8, st));
        unsigned char *bounce = nullptr;
        if (vmm == 3) {
            CK(hipMalloc((void **)&bounce, n));
            CK(hipMemcpyAsync(bounce, host.data(), n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, st, dst, (const unsigned char *)bounce, n);
        } else CK(hipMemcpyAsync(dst, host.data(), n, hipMemcpyHostToDevice, st));          // pageable source, stream-ordered
        if (sync_after_copy) CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, dst, n, dsum);
        unsigned long long got = 0; CK(hipMemcpyAsync(&got, dsum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        if (bounce) CK(hipFree(bounce));
        if (got != want) { if (++badv[vmm] <= 3) printf("%s n=%zu rep %d: kernel saw %llu, expected %llu\n", vmm == 0 ? "malloc" : vmm == 1 ? "VMM stream-ordered" : vmm == 2 ? "VMM host waits" : "VMM via bounce", n, rep, got, want); }
        if (!vmm) CK(hipFree(dst)); else { CK(hipMemUnmap((char *)va + gran, mapped)); CK(hipMemRelease(h)); CK(hipMemAddressFree(va, mapped + 2 * gran)); }
    }
    printf("vmm_h2d_repro: mismatches of 80 copies each — hipMalloc %d, VMM stream-ordered %d, VMM with the host waiting %d, VMM through a bounce buffer + copy kernel %d\n", badv[0], badv[1], badv[2], badv[3]);
    return badv[3] || badv[0] ? 1 : 0;
}
