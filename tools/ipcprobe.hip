// ipcprobe.hip — what the position-sharded chain needs from the platform, measured with G processes on ONE GPU:
//   1. hipIpcGetMemHandle / hipIpcOpenMemHandle on plain hipMalloc memory and on hipExtMallocWithFlags(uncached / fine-grained)
//   2. do kernels of different processes run CONCURRENTLY (a kernel spinning on a flag another process's kernel sets)?
//   3. latency of a cross-process flag barrier done by one tiny kernel per rank (signal all peers, wait for all peers)
//   4. peer stores into a neighbour's coarse-grained buffer + barrier kernel + read-back in the next kernel: is the data there?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ipcprobe tools/ipcprobe.hip ; run: tools/ipcprobe [G=2] [iters=2000] [only: memory kind 0..2, 3 = VMM]
// IPCPROBE_SPREAD=1 puts rank r on device r (tests/test_gpu_multi.py runs `ipcprobe <world> 200 0` that way before the position-sharded check)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <sys/socket.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[r%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); _exit(3); } } while (0)
static int g_rank = -1;
constexpr int MAXG = 8;

struct Shared {                         // host shared memory between the processes
    volatile int arrive[64];
    hipIpcMemHandle_t hFlags[MAXG], hData[MAXG];
    volatile int ok[MAXG];
};
static void host_barrier(Shared *sh, int G, int &phase) {
    __sync_fetch_and_add(&sh->arrive[phase], 1);
    while (sh->arrive[phase] < G) usleep(50);
    ++phase;
}

struct Peers { unsigned *flags[MAXG]; int *data[MAXG]; };

// signal every peer (flags[p][me] = epoch), then wait until every peer has signalled me; bounded spin
__global__ void xbar_kernel(Peers P, int me, int G, unsigned epoch, int *err) {
    const int t = threadIdx.x;
    if (t < G) {
        __hip_atomic_store(P.flags[t] + me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        long spins = 0;
        while ((int)(__hip_atomic_load(P.flags[me] + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 24)) { atomicExch(err, 1 + t); break; }
        }
    }
}
// every rank stores a pattern into the NEXT rank's buffer (peer store through the IPC mapping)
__global__ void scatter_kernel(Peers P, int me, int G, int n, int it) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) P.data[(me + 1) % G][(int)(((unsigned)i * 7919u) % (unsigned)n)] = it * 1000003 + i + me;       // scattered 4-byte stores
}
__global__ void read_kernel(const int *p, int *sink) { if (p[threadIdx.x] == 0x7fffffff) atomicAdd(sink, 1); }
__global__ void write_kernel(int *p) { p[threadIdx.x] = 7; }
__global__ void check_kernel(const int *mine, int from, int n, int it, int *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mine[(int)(((unsigned)i * 7919u) % (unsigned)n)] != it * 1000003 + i + from) atomicAdd(bad, 1);
}

// IPCPROBE_SPREAD=1: rank r on device r mod device count (a multi-GPU node: the peer stores, flags and read-backs of run_rank cross xGMI); default: one GPU
static int probe_device(int rank) {
    const char *s = getenv("IPCPROBE_SPREAD");
    int n = 1;
    if (!s || !atoi(s) || hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
    return rank % n;
}
static int run_rank(Shared *sh, int rank, int G, int iters, int kind) {
    g_rank = rank;
    setvbuf(stdout, nullptr, _IONBF, 0);
    int phase = kind * 16;
    CK(hipSetDevice(probe_device(rank)));
    const int n = 1 << 20;
    unsigned *flags = nullptr; int *data = nullptr, *err = nullptr, *bad = nullptr;
    if (kind == 0) CK(hipMalloc((void **)&flags, 4096));
    else {
        hipError_t e = hipExtMallocWithFlags((void **)&flags, 4096, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("[r%d] kind %d: hipExtMallocWithFlags -> %s\n", rank, kind, hipGetErrorString(e)); (void)hipGetLastError(); sh->ok[rank] = 0; flags = nullptr; }
    }
    if (flags) sh->ok[rank] = 1;
    host_barrier(sh, G, phase);
    for (int r = 0; r < G; ++r) if (!sh->ok[r]) { if (rank == 0) printf("kind %d: allocation failed on some rank, skipped\n", kind); return 0; }
    CK(hipMalloc((void **)&data, n * sizeof(int)));
    CK(hipMalloc((void **)&err, 8)); CK(hipMalloc((void **)&bad, 8));
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(err, 0, 8)); CK(hipMemset(bad, 0, 8)); CK(hipMemset(data, 0, n * sizeof(int)));
    CK(hipDeviceSynchronize());
    hipError_t e1 = hipIpcGetMemHandle(&sh->hFlags[rank], flags);
    if (e1 != hipSuccess) { printf("[r%d] kind %d: hipIpcGetMemHandle(flags) -> %s\n", rank, kind, hipGetErrorString(e1)); sh->ok[rank] = 0; (void)hipGetLastError(); }
    CK(hipIpcGetMemHandle(&sh->hData[rank], data));
    host_barrier(sh, G, phase);
    for (int r = 0; r < G; ++r) if (!sh->ok[r]) { if (rank == 0) printf("kind %d: IPC handle of this memory kind refused, skipped\n", kind); return 0; }
    Peers P; memset(&P, 0, sizeof P);
    for (int r = 0; r < G; ++r) {
        if (r == rank) { P.flags[r] = flags; P.data[r] = data; continue; }
        CK(hipIpcOpenMemHandle((void **)&P.flags[r], sh->hFlags[r], hipIpcMemLazyEnablePeerAccess));
        CK(hipIpcOpenMemHandle((void **)&P.data[r], sh->hData[r], hipIpcMemLazyEnablePeerAccess));
    }
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    host_barrier(sh, G, phase);
    {   // 1b. which accesses does the imported mapping allow?  (each step printed before the next is tried)
        const int nx = (rank + 1) % G;
        int v = -1;
        printf("[r%d] kind %d: own flags %p data %p; peer %d flags %p data %p\n", rank, kind, (void *)flags, (void *)data, nx, (void *)P.flags[nx], (void *)P.data[nx]);
        CK(hipMemcpy(&v, P.data[nx], 4, hipMemcpyDeviceToHost)); printf("[r%d] kind %d: hipMemcpy D2H from the peer mapping ok (%d)\n", rank, kind, v);
        hipLaunchKernelGGL(read_kernel, dim3(1), dim3(64), 0, st, (const int *)P.data[nx], bad); CK(hipStreamSynchronize(st)); printf("[r%d] kind %d: kernel READ of peer data ok\n", rank, kind);
        hipLaunchKernelGGL(read_kernel, dim3(1), dim3(64), 0, st, (const int *)P.flags[nx], bad); CK(hipStreamSynchronize(st)); printf("[r%d] kind %d: kernel READ of peer flags ok\n", rank, kind);
        v = 5; CK(hipMemcpy(P.data[nx] + 100, &v, 4, hipMemcpyHostToDevice)); printf("[r%d] kind %d: hipMemcpy H2D into the peer mapping ok\n", rank, kind);
        hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, st, P.data[nx] + 256); CK(hipStreamSynchronize(st)); printf("[r%d] kind %d: kernel WRITE of peer data ok\n", rank, kind);
        hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, st, (int *)P.flags[nx] + 256); CK(hipStreamSynchronize(st)); printf("[r%d] kind %d: kernel WRITE of peer flags ok\n", rank, kind);
        CK(hipMemset(bad, 0, 8));
    }
    host_barrier(sh, G, phase);
    // 2. concurrency: one barrier with a deliberate 200 ms stagger — rank 0's kernel must sit spinning while rank 1 has not launched yet
    unsigned epoch = 1;
    if (rank != 0) usleep(200000);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(xbar_kernel, dim3(1), dim3(64), 0, st, P, rank, G, epoch, err);
    CK(hipStreamSynchronize(st));
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("[r%d] kind %d staggered barrier: %.1f ms, err %d (%s)\n", rank, kind, ms, herr, herr ? "TIMEOUT: kernels of two processes did not overlap" : "concurrent");
    host_barrier(sh, G, phase);
    if (herr) return 1;
    // 3. barrier latency, back to back
    host_barrier(sh, G, phase);
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(xbar_kernel, dim3(1), dim3(64), 0, st, P, rank, G, ++epoch, err);
    CK(hipStreamSynchronize(st));
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("[r%d] kind %d: %d barriers %.2f us each, err %d\n", rank, kind, iters, ms * 1000 / iters, herr);
    host_barrier(sh, G, phase);
    // 4. peer scatter -> barrier -> check -> barrier, iterated
    t0 = std::chrono::steady_clock::now();
    const int its = iters / 4;
    for (int it = 1; it <= its; ++it) {
        hipLaunchKernelGGL(scatter_kernel, dim3(n / 256), dim3(256), 0, st, P, rank, G, n, it);
        hipLaunchKernelGGL(xbar_kernel, dim3(1), dim3(64), 0, st, P, rank, G, ++epoch, err);
        hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, st, (const int *)data, (rank + G - 1) % G, n, it, bad);
        hipLaunchKernelGGL(xbar_kernel, dim3(1), dim3(64), 0, st, P, rank, G, ++epoch, err);
    }
    CK(hipStreamSynchronize(st));
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int hbad = 0; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("[r%d] kind %d: %d x (4 MB peer scatter, barrier, check, barrier): %.2f us per iteration, mismatches %d, err %d\n", rank, kind, its, ms * 1000 / its, hbad, herr);
    host_barrier(sh, G, phase);
    for (int r = 0; r < G; ++r) if (r != rank) { (void)hipIpcCloseMemHandle(P.flags[r]); (void)hipIpcCloseMemHandle(P.data[r]); }
    host_barrier(sh, G, phase);
    (void)hipFree(flags); (void)hipFree(data);
    return (hbad || herr) ? 1 : 0;
}


// ---- the virtual-memory route: hipMemCreate (shareable POSIX fd) -> fd over a unix socket -> hipMemImportFromShareableHandle -> hipMemMap
static int send_fd(int sock, int fd) {
    char b = 'x'; struct iovec io = {&b, 1}; char cb[CMSG_SPACE(sizeof(int))]; memset(cb, 0, sizeof cb);
    struct msghdr m = {}; m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = cb; m.msg_controllen = sizeof cb;
    struct cmsghdr *c = CMSG_FIRSTHDR(&m); c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    return sendmsg(sock, &m, 0) == 1 ? 0 : -1;
}
static int recv_fd(int sock) {
    char b; struct iovec io = {&b, 1}; char cb[CMSG_SPACE(sizeof(int))];
    struct msghdr m = {}; m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = cb; m.msg_controllen = sizeof cb;
    if (recvmsg(sock, &m, 0) != 1) return -1;
    struct cmsghdr *c = CMSG_FIRSTHDR(&m); int fd = -1; if (c) memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}
static int run_vmm(Shared *sh, int rank, int G, int (*chan)[2]) {
    int phase = 48;
    const int nx = (rank + 1) % G, pv = (rank + G - 1) % G;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t sz = ((4u << 20) + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0));
    void *mine = nullptr; CK(hipMemAddressReserve(&mine, sz, gran, nullptr, 0)); CK(hipMemMap(mine, sz, 0, h, 0));
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(mine, sz, &ad, 1));
    int fd = -1; CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
    printf("[r%d] vmm: exported fd %d (granularity %zu)\n", rank, fd, gran);
    // rank r's buffer is written by rank r-1: send my fd to pv over chan[pv] (pv receives on [0], I send on [1])
    if (send_fd(chan[pv][1], fd)) { printf("[r%d] vmm: send_fd failed\n", rank); return 1; }
    const int pfd = recv_fd(chan[rank][0]);
    if (pfd < 0) { printf("[r%d] vmm: recv_fd failed\n", rank); return 1; }
    hipMemGenericAllocationHandle_t ph;
    hipError_t e = hipMemImportFromShareableHandle(&ph, (void *)(uintptr_t)pfd, hipMemHandleTypePosixFileDescriptor);
    if (e != hipSuccess) { printf("[r%d] vmm: import -> %s; retrying with a pointer to the fd\n", rank, hipGetErrorString(e)); (void)hipGetLastError(); int tmp = pfd; CK(hipMemImportFromShareableHandle(&ph, (void *)&tmp, hipMemHandleTypePosixFileDescriptor)); }
    void *peer = nullptr; CK(hipMemAddressReserve(&peer, sz, gran, nullptr, 0)); CK(hipMemMap(peer, sz, 0, ph, 0)); CK(hipMemSetAccess(peer, sz, &ad, 1));
    printf("[r%d] vmm: own %p, peer %d mapped at %p\n", rank, mine, nx, peer);
    CK(hipMemset(mine, 0, sz)); CK(hipDeviceSynchronize());
    host_barrier(sh, G, phase);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *bad = nullptr; CK(hipMalloc((void **)&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, st, (int *)peer + 256); CK(hipStreamSynchronize(st));
    printf("[r%d] vmm: kernel WRITE into the peer's memory ok\n", rank);
    host_barrier(sh, G, phase);
    int v[2] = {0, 0}; CK(hipMemcpy(v, (int *)mine + 256, 8, hipMemcpyDeviceToHost));
    printf("[r%d] vmm: my buffer holds %d %d after the neighbour's write (want 7 7)\n", rank, v[0], v[1]);
    host_barrier(sh, G, phase);
    return (v[0] == 7 && v[1] == 7) ? 0 : 1;
}

// ---- 5. how large may an exported allocation be?  (the ring of a 1 M-haplotype engine is > 4 GiB)
static int run_big(Shared *sh, int rank, int G) {
    int phase = 56;
    const double gbs[] = {1.0, 2.0, 3.9, 4.1, 6.0};
    for (double gb : gbs) {
        const size_t n = (size_t)(gb * (1ull << 30));
        char *buf = nullptr; CK(hipMalloc((void **)&buf, n)); CK(hipMemset(buf, 0, 4096)); CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        CK(hipIpcGetMemHandle(&sh->hData[rank], buf));
        double msg = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        host_barrier(sh, G, phase);
        const int nx = (rank + 1) % G;
        void *peer = nullptr;
        t0 = std::chrono::steady_clock::now();
        hipError_t e = hipIpcOpenMemHandle(&peer, sh->hData[nx], hipIpcMemLazyEnablePeerAccess);
        double mso = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("[r%d] %.1f GiB: get handle %.1f ms, open %.1f ms -> %s\n", rank, gb, msg, mso, hipGetErrorString(e));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, 0, (int *)((char *)peer + n - 4096)); CK(hipDeviceSynchronize());
            printf("[r%d] %.1f GiB: kernel write at the end of the peer mapping ok\n", rank, gb);
            (void)hipIpcCloseMemHandle(peer);
        } else (void)hipGetLastError();
        host_barrier(sh, G, phase);
        CK(hipFree(buf));
        host_barrier(sh, G, phase);
    }
    return 0;
}

// ---- 6. several imported buffers held open at once: is there a limit on the TOTAL imported size?
static int run_cumulative(Shared *sh, int rank, int G) {
    int phase = 40;
    const int NB = 5; const size_t n = 2ull << 30;
    char *buf[NB]; void *peer[NB];
    static hipIpcMemHandle_t *hs = nullptr;
    for (int i = 0; i < NB; ++i) { CK(hipMalloc((void **)&buf[i], n)); CK(hipMemset(buf[i], 0, 4096)); }
    CK(hipDeviceSynchronize());
    const int nx = (rank + 1) % G;
    for (int i = 0; i < NB; ++i) {
        CK(hipIpcGetMemHandle(&sh->hData[rank], buf[i]));
        host_barrier(sh, G, phase);
        auto t0 = std::chrono::steady_clock::now();
        printf("[r%d] opening buffer %d (%d GiB imported so far)\n", rank, i, 2 * i);
        hipError_t e = hipIpcOpenMemHandle(&peer[i], sh->hData[nx], hipIpcMemLazyEnablePeerAccess);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("[r%d] buffer %d open: %.1f ms -> %s\n", rank, i, ms, hipGetErrorString(e));
        if (e != hipSuccess) return 1;
        hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, 0, (int *)((char *)peer[i] + n - 4096)); CK(hipDeviceSynchronize());
        host_barrier(sh, G, phase);
    }
    (void)hs;
    return 0;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 2000;
    if (G < 2 || G > MAXG) return 2;
    Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset(sh, 0, sizeof *sh);
    pid_t pids[MAXG];
    int chan[MAXG][2];
    for (int r = 0; r < G; ++r) if (socketpair(AF_UNIX, SOCK_STREAM, 0, chan[r])) return 2;
    const int only = argc > 3 ? atoi(argv[3]) : -1;          // run one memory kind only (0..2), 3 = the VMM route only
    for (int r = 0; r < G; ++r) {
        pids[r] = fork();                                   // before any HIP call: every rank initialises its own runtime
        if (pids[r] == 0) {
            int rc = 0;
            g_rank = r; setvbuf(stdout, nullptr, _IONBF, 0);
            if (only == 4) { CK(hipSetDevice(0)); rc |= run_big(sh, r, G); }
            if (only == 5) { CK(hipSetDevice(0)); rc |= run_cumulative(sh, r, G); }
            if (only == 3 || only < 0) { CK(hipSetDevice(0)); rc |= run_vmm(sh, r, G, chan); }
            for (int kind = 0; kind < 3; ++kind) if (only < 0 || only == kind) rc |= run_rank(sh, r, G, iters, kind);
            fflush(stdout);
            _exit(rc);
        }
    }
    int bad = 0;
    for (int r = 0; r < G; ++r) { int st = 0; waitpid(pids[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) bad = 1; }
    printf("ipcprobe: %s\n", bad ? "FAILED" : "ok");
    return bad;
}
