// tools/mallprobe.hip — does the 256 MiB Infinity Cache keep what a kernel has just WRITTEN for the next kernel's reads?
// (the fill -> sweep hand-off of DESIGN.md section 4: the fill stores d | y of a batch's states, the sweep reads them back).
// For a working set of S bytes: kernel W stores S bytes (plain or nontemporal), kernel R reads them back; both timed with events, over a
// ring of distinct regions so that nothing is re-used by accident; the figure of interest is R's rate against S.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int NT> __global__ __launch_bounds__(256) void wr(int4 *p, size_t n, int v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int4 x = make_int4(v, (int)i, v, v);
        if (NT) __builtin_nontemporal_store(x.x, &p[i].x), __builtin_nontemporal_store(x.y, &p[i].y), __builtin_nontemporal_store(x.z, &p[i].z), __builtin_nontemporal_store(x.w, &p[i].w);
        else p[i] = x;
    }
}
template <int NT> __global__ __launch_bounds__(256) void rd(const int4 *p, size_t n, int *out) {
    int s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int4 x;
        if (NT) { x.x = __builtin_nontemporal_load(&p[i].x); x.y = __builtin_nontemporal_load(&p[i].y); x.z = __builtin_nontemporal_load(&p[i].z); x.w = __builtin_nontemporal_load(&p[i].w); }
        else x = p[i];
        s += x.x ^ x.y ^ x.z ^ x.w;
    }
    if (s == 0x12345678) *out = s;
}
int main() {
    const size_t TOT = (size_t)4 << 30;
    int4 *buf; int *out; CHK(hipMalloc(&buf, TOT)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(buf, 0, TOT));
    hipEvent_t e0, e1, e2; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2));
    printf("%10s %4s %4s | write GB/s | read-back GB/s\n", "S (MiB)", "w nt", "r nt");
    for (int wnt = 0; wnt < 2; ++wnt) for (int rnt = 0; rnt < 2; ++rnt)
    for (size_t S = (size_t)16 << 20; S <= ((size_t)2 << 30); S <<= 1) {
        const size_t n = S / 16, regions = TOT / S;
        double tw = 0, tr = 0; int cnt = 0;
        const size_t use = regions < 16 ? regions : 16;
        for (int rep = 0; rep < 3; ++rep)
            for (size_t r = 0; r < use; ++r) {
                int4 *p = buf + r * n;
                CHK(hipEventRecord(e0));
                if (wnt) wr<1><<<2048, 256>>>(p, n, rep); else wr<0><<<2048, 256>>>(p, n, rep);
                CHK(hipEventRecord(e1));
                if (rnt) rd<1><<<2048, 256>>>(p, n, out); else rd<0><<<2048, 256>>>(p, n, out);
                CHK(hipEventRecord(e2)); CHK(hipEventSynchronize(e2));
                float a, b; CHK(hipEventElapsedTime(&a, e0, e1)); CHK(hipEventElapsedTime(&b, e1, e2));
                if (rep) { tw += a; tr += b; ++cnt; }
            }
        printf("%10zu %4d %4d | %10.0f | %10.0f\n", S >> 20, wnt, rnt, S / 1e6 / (tw / cnt), S / 1e6 / (tr / cnt));
    }
    return 0;
}
