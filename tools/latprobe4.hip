// latprobe4.hip — what a dependency costs INSIDE ONE XCD.
// DESIGN.md section 2 prices a kernel boundary (1.54 us + ~1.0 us first load) and a barrier over all 8 XCDs (>= 4 us).  Not priced so far:
// a barrier among the <= 32 CUs of one XCD, whose L2 (4 MB) is the coherence point of everything they exchange — arrivals as L2-scope
// atomics (no sc1), payload as plain stores that stay in that L2, readers bypassing their L1 (sc1 / nt loads) instead of fencing.
//   (1) census: which XCD a CU-mask bit selects (hipExtStreamCreateWithCUMask), read back with s_getreg(HW_REG_XCC_ID);
//   (2) barrier only: K workgroups on one XCD, R rounds, by arrival scope;
//   (3) produce 1 MB -> barrier -> consume 1 MB (another workgroup's slice, checked word by word) by load flavour;
//   (4) the same phases as dependent launches on the same masked stream (what the persistent form has to beat).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }     // HW_REG_XCC_ID[3:0]

__global__ void census_kernel(int *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// scope 0: agent-scope arrive + agent-scope poll; 1: workgroup-scope arrive (executes in the XCD's L2, no sc1) + sc1 poll;
// 2: workgroup-scope arrive, poll with a workgroup-scope RMW (add 0); 3: NO atomics — every workgroup stores the round into its own flag word
// (plain store: lands in the XCD's L2), wave 0 polls all flags with sc1 loads (L1-bypassing, L2-served)
__shared__ int s_abort;                                    // set by the polling wave on a timeout (or when another workgroup timed out): uniform exit
template <int SCOPE>
__device__ __forceinline__ void xbarrier(unsigned *counter, unsigned target, int *err) {
    __syncthreads();
    if (SCOPE == 3) {
        if (threadIdx.x < 64) {
            const unsigned round = target / gridDim.x;         // target = base + (r + 1) * nwg, base a multiple of nwg
            if (threadIdx.x == 0) __hip_atomic_store(counter + 64 + blockIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) ok &= (int)(__hip_atomic_load(counter + 64 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - round) >= 0;
                if (__all(ok)) break;
                if (++spins > (1 << 19) || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { atomicExch(err, 1); s_abort = 1; break; }
            }
        }
    } else if (threadIdx.x == 0) {
        if (SCOPE == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        for (;;) {
            unsigned v;
            if (SCOPE == 2) v = __hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) break;
            if (++spins > (1 << 19) || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { atomicExch(err, 1); s_abort = 1; break; }
        }
    }
    __syncthreads();
}

template <int SCOPE>
__global__ __launch_bounds__(256) void barrier_only_kernel(unsigned *counter, unsigned base, int rounds, int *err, int *xcc_seen) {
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1 << xcc_id());
    unsigned target = base;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) { target += gridDim.x; xbarrier<SCOPE>(counter, target, err); if (s_abort) break; }
}

typedef int v4i __attribute__((ext_vector_type(4)));
// LOADF 0: plain, 1: nontemporal, 2: 8-byte agent-scope relaxed atomic loads (sc1), 3: plain after an agent acquire fence (buffer_inv sc1),
// 4: 16-byte sc1 loads (inline asm)
template <int SCOPE, int LOADF>
__global__ __launch_bounds__(256) void prodcons_kernel(int *buf0, int *buf1, int words_per_wg, unsigned *counter, unsigned base, int rounds, int *err, unsigned long long *bad) {
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
    unsigned target = base;
    unsigned long long nbad = 0;
    if (t == 0) s_abort = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        if (s_abort) break;
        int *buf = (r & 1) ? buf1 : buf0;
        int *mine = buf + (size_t)b * words_per_wg;
        for (int i = t * 4; i < words_per_wg; i += 1024) {
            const int tag = (r << 20) ^ (b * words_per_wg + i);
            *reinterpret_cast<v4i *>(mine + i) = v4i{tag, tag + 1, tag + 2, tag + 3};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        target += nb; xbarrier<SCOPE>(counter, target, err);
        const int src = (b + nb / 2 + 1) % nb;
        const int *theirs = buf + (size_t)src * words_per_wg;
        if (LOADF == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = t * 4; i < words_per_wg; i += 1024) {
            const int tag = (r << 20) ^ (src * words_per_wg + i);
            v4i v;
            if (LOADF == 0 || LOADF == 3) v = *reinterpret_cast<const v4i *>(theirs + i);
            else if (LOADF == 1) v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(theirs + i));
            else if (LOADF == 2) {
                const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(theirs + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(theirs + i + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = v4i{(int)lo, (int)(lo >> 32), (int)hi, (int)(hi >> 32)};
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
            }
            nbad += (v.x != tag) + (v.y != tag + 1) + (v.z != tag + 2) + (v.w != tag + 3);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void produce_kernel(int *buf, int words_per_wg, int r) {
    const int t = threadIdx.x, b = blockIdx.x;
    int *mine = buf + (size_t)b * words_per_wg;
    for (int i = t * 4; i < words_per_wg; i += 1024) { const int tag = (r << 20) ^ (b * words_per_wg + i); *reinterpret_cast<v4i *>(mine + i) = v4i{tag, tag + 1, tag + 2, tag + 3}; }
}
__global__ __launch_bounds__(256) void consume_kernel(const int *buf, int words_per_wg, int r, unsigned long long *bad) {
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x, src = (b + nb / 2 + 1) % nb;
    const int *theirs = buf + (size_t)src * words_per_wg;
    unsigned long long nbad = 0;
    for (int i = t * 4; i < words_per_wg; i += 1024) { const int tag = (r << 20) ^ (src * words_per_wg + i); const v4i v = *reinterpret_cast<const v4i *>(theirs + i); nbad += (v.x != tag) + (v.y != tag + 1) + (v.z != tag + 2) + (v.w != tag + 3); }
    if (nbad) atomicAdd(bad, nbad);
}
// produce + consume in one kernel (read the previous launch's buffer, write this launch's): one boundary per round, as the chain's launches are
__global__ __launch_bounds__(256) void prodcons_launch_kernel(const int *bin, int *bout, int words_per_wg, int r, unsigned long long *bad) {
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x, src = (b + nb / 2 + 1) % nb;
    unsigned long long nbad = 0;
    for (int i = t * 4; i < words_per_wg; i += 1024) {
        const int tag = ((r - 1) << 20) ^ (src * words_per_wg + i); const v4i v = *reinterpret_cast<const v4i *>(bin + (size_t)src * words_per_wg + i);
        nbad += (v.x != tag) + (v.y != tag + 1) + (v.z != tag + 2) + (v.w != tag + 3);
        const int tg = (r << 20) ^ (b * words_per_wg + i);
        *reinterpret_cast<v4i *>(bout + (size_t)b * words_per_wg + i) = v4i{tg, tg + 1, tg + 2, tg + 3};
    }
    if (nbad && r > 0) atomicAdd(bad, nbad);
}


// (5) a TEAM on ONE XCD chosen at run time: the launch has 8 K workgroups, a workgroup reads HW_REG_XCC_ID and leaves unless it is on XCD `xcd`;
// the K that stay (block b runs on XCD b % 8: observed, checked by the barrier itself — it times out if fewer arrive) take ranks from a ticket
// counter and synchronise through flag words with NO far atomics: arrive = plain store of the round into the member's own word (lands in
// this XCD's L2, the coherence point of every team member), poll = sc1 loads of all K words by wave 0 (L1-bypassing, served by that L2).
// Payload: plain stores, read back by another member with the load flavour LOADF.  payload_words == 0: barrier only.
template <int LOADF>
__global__ __launch_bounds__(256) void team_kernel(int xcd, int K, int *buf0, int *buf1, int words_per_wg, unsigned *flags, unsigned *ticket, unsigned round0, int rounds, int *err, unsigned long long *bad) {
    __shared__ int s_rank;
    const int t = threadIdx.x;
    if (xcc_id() != xcd) return;
    if (t == 0) s_rank = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) s_abort = 0;
    __syncthreads();
    const int b = s_rank;
    if (b >= K) return;
    unsigned long long nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (s_abort) break;
        int *buf = (r & 1) ? buf1 : buf0;
        for (int i = t * 4; i < words_per_wg; i += 1024) {
            const int tag = (r << 20) ^ (b * words_per_wg + i);
            *reinterpret_cast<v4i *>(buf + (size_t)b * words_per_wg + i) = v4i{tag, tag + 1, tag + 2, tag + 3};
        }
        __syncthreads();                                    // every wave's stores acknowledged by L2 (vmcnt(0)) before the flag
        if (t < 64) {
            const unsigned round = round0 + (unsigned)r + 1u;
            if (t == 0) __hip_atomic_store(flags + b, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (int i = t; i < K; i += 64) ok &= (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - round) >= 0;
                if (__all(ok)) break;
                if (++spins > (1 << 19)) { atomicExch(err, 1); s_abort = 1; break; }
            }
        }
        __syncthreads();
        const int src = (b + K / 2 + 1) % K;
        const int *theirs = buf + (size_t)src * words_per_wg;
        for (int i = t * 4; i < words_per_wg; i += 1024) {
            const int tag = (r << 20) ^ (src * words_per_wg + i);
            v4i v;
            if (LOADF == 0) v = *reinterpret_cast<const v4i *>(theirs + i);
            else if (LOADF == 1) v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(theirs + i));
            else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
            nbad += (v.x != tag) + (v.y != tag + 1) + (v.z != tag + 2) + (v.w != tag + 3);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

static hipStream_t masked_stream(int xcd, bool interleaved) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) { const bool on = interleaved ? (i % 8 == xcd) : (i / 32 == xcd); if (on) mask[i / 32] |= 1u << (i % 32); }
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return st;
}

struct Timer { hipEvent_t a, b; Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); } };

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int *d_census, *d_err, *d_xcc; unsigned *d_counter; unsigned long long *d_bad;
    CK(hipMalloc(&d_census, 4096 * 4)); CK(hipMalloc(&d_err, 4)); CK(hipMalloc(&d_xcc, 4)); CK(hipMalloc(&d_counter, 8192)); CK(hipMalloc(&d_bad, 8));
    CK(hipMemset(d_err, 0, 4)); CK(hipMemset(d_counter, 0, 8192)); CK(hipMemset(d_bad, 0, 8));
    std::vector<int> h(4096);
    // (1) census
    for (int form = 0; form < 3; ++form) {
        hipStream_t st = form == 0 ? nullptr : masked_stream(3, form == 1);
        if (form && !st) { printf("census form %d: no masked stream\n", form); continue; }
        CK(hipMemset(d_census, 0xff, 4096 * 4));
        hipLaunchKernelGGL(census_kernel, dim3(1024), dim3(64), 0, st, d_census);
        CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), d_census, 1024 * 4, hipMemcpyDeviceToHost));
        int cnt[16] = {0}, rr = 0; for (int i = 0; i < 1024; ++i) { cnt[h[i] & 15]++; rr += (h[i] == (i & 7)); }
        printf("census %-28s: blocks per XCC id:", form == 0 ? "no mask" : form == 1 ? "mask bits i%8==3" : "mask bits i/32==3");
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("   (block b on XCC b%%8: %d of 1024)\n", rr);
        if (st) CK(hipStreamDestroy(st));
    }
    hipStream_t sx = masked_stream(0, true);
    if (!sx) { printf("no CU-masked stream: stopping\n"); return 0; }
    Timer T; float ms;
    unsigned base = 0;
    auto run_barrier = [&](const char *name, hipStream_t st, int nwg, int scope, int rounds) {
        CK(hipMemset(d_xcc, 0, 4)); CK(hipMemset(d_counter, 0, 8192)); base = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(T.a, st));
            if (scope == 0) hipLaunchKernelGGL(barrier_only_kernel<0>, dim3(nwg), dim3(256), 0, st, d_counter, base, rounds, d_err, d_xcc);
            else if (scope == 1) hipLaunchKernelGGL(barrier_only_kernel<1>, dim3(nwg), dim3(256), 0, st, d_counter, base, rounds, d_err, d_xcc);
            else if (scope == 2) hipLaunchKernelGGL(barrier_only_kernel<2>, dim3(nwg), dim3(256), 0, st, d_counter, base, rounds, d_err, d_xcc);
            else hipLaunchKernelGGL(barrier_only_kernel<3>, dim3(nwg), dim3(256), 0, st, d_counter, base, rounds, d_err, d_xcc);
            CK(hipEventRecord(T.b, st)); CK(hipStreamSynchronize(st)); base += (unsigned)nwg * rounds;
        }
        CK(hipEventElapsedTime(&ms, T.a, T.b));
        int err = 0, xm = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&xm, d_xcc, 4, hipMemcpyDeviceToHost));
        printf("barrier only   %-34s %3d WGs: %.3f us per barrier   (XCC mask seen 0x%02x%s)\n", name, nwg, ms * 1e3 / rounds, xm, err ? ", TIMEOUT" : "");
        if (err) CK(hipMemset(d_err, 0, 4));
    };
    const int R = 4000;
    for (int nwg : {8, 32, 64, 128}) {
        run_barrier("one XCD, agent-scope arrive+poll", sx, nwg, 0, R);
        run_barrier("one XCD, L2-scope arrive, sc1 poll", sx, nwg, 1, R);
        run_barrier("one XCD, L2-scope arrive + RMW poll", sx, nwg, 2, R);
        run_barrier("one XCD, flag words, no atomics", sx, nwg, 3, R);
    }
    for (int nwg : {64, 256, 512}) run_barrier("all XCDs, agent-scope", nullptr, nwg, 0, R);
    // (3) produce -> barrier -> consume
    const size_t total_words = 256 * 1024;                  // 1 MB
    int *b0, *b1; CK(hipMalloc(&b0, total_words * 4 * 2)); b1 = b0 + total_words; CK(hipMemset(b0, 0, total_words * 8));
    auto run_pc = [&](const char *name, hipStream_t st, int nwg, int scope, int loadf, int rounds, size_t words) {
        const int wpw = (int)(words / nwg);
        CK(hipMemset(d_counter, 0, 8192)); CK(hipMemset(d_bad, 0, 8)); base = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipMemset(d_bad, 0, 8));
            CK(hipEventRecord(T.a, st));
#define PC(S, L) hipLaunchKernelGGL((prodcons_kernel<S, L>), dim3(nwg), dim3(256), 0, st, b0, b1, wpw, d_counter, base, rounds, d_err, d_bad)
            if (scope == 0) { if (loadf == 0) PC(0, 0); else if (loadf == 1) PC(0, 1); else if (loadf == 2) PC(0, 2); else if (loadf == 3) PC(0, 3); else PC(0, 4); }
            else if (scope == 3) { if (loadf == 0) PC(3, 0); else if (loadf == 1) PC(3, 1); else if (loadf == 2) PC(3, 2); else if (loadf == 3) PC(3, 3); else PC(3, 4); }
            else { if (loadf == 0) PC(1, 0); else if (loadf == 1) PC(1, 1); else if (loadf == 2) PC(1, 2); else if (loadf == 3) PC(1, 3); else PC(1, 4); }
#undef PC
            CK(hipEventRecord(T.b, st)); CK(hipStreamSynchronize(st)); base += (unsigned)nwg * rounds;
        }
        CK(hipEventElapsedTime(&ms, T.a, T.b));
        unsigned long long bad = 0; int err = 0; CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("produce->barrier->consume %4zu KB %-30s %3d WGs: %.3f us per round, stale words %llu of %llu%s\n", words * 4 / 1024, name, nwg, ms * 1e3 / rounds,
               bad, (unsigned long long)words * rounds, err ? ", TIMEOUT" : "");
        if (err) CK(hipMemset(d_err, 0, 4));
    };
    const char *lname[5] = {"plain loads", "nt loads", "8-B sc1 atomic loads", "acquire fence + plain", "16-B sc1 loads (asm)"};
    for (size_t words : {total_words, total_words / 4})
        for (int nwg : {32, 64, 128})
            for (int lf = 0; lf < 5; ++lf) { char nm[96]; snprintf(nm, sizeof nm, "L2-scope, %s", lname[lf]); run_pc(nm, sx, nwg, 1, lf, 2000, words); }
    for (int lf : {1, 4}) { char nm[96]; snprintf(nm, sizeof nm, "agent-scope, %s", lname[lf]); run_pc(nm, sx, 64, 0, lf, 2000, total_words); }
    for (size_t words : {total_words, total_words / 4})
        for (int nwg : {32, 64, 128})
            for (int lf : {1, 2, 4}) { char nm[96]; snprintf(nm, sizeof nm, "flag words, %s", lname[lf]); run_pc(nm, sx, nwg, 3, lf, 2000, words); }
    // (4) the same as dependent launches on the masked stream, and on the whole chip
    for (int whole = 0; whole < 2; ++whole) {
        hipStream_t st = whole ? nullptr : sx;
        for (size_t words : {total_words, total_words / 4})
            for (int nwg : {32, 128, 512}) {
                const int wpw = (int)(words / nwg), rounds = 2000;
                CK(hipMemset(d_bad, 0, 8));
                for (int r = 0; r < 16; ++r) hipLaunchKernelGGL(prodcons_launch_kernel, dim3(nwg), dim3(256), 0, st, (r & 1) ? b0 : b1, (r & 1) ? b1 : b0, wpw, 0, d_bad);
                CK(hipStreamSynchronize(st)); CK(hipMemset(d_bad, 0, 8));
                CK(hipEventRecord(T.a, st));
                for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(prodcons_launch_kernel, dim3(nwg), dim3(256), 0, st, (r & 1) ? b0 : b1, (r & 1) ? b1 : b0, wpw, r, d_bad);
                CK(hipEventRecord(T.b, st)); CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, T.a, T.b));
                unsigned long long bad = 0; CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
                printf("dependent launches %4zu KB, %s, %3d WGs: %.3f us per launch (stale %llu)\n", words * 4 / 1024, whole ? "whole chip" : "one XCD   ", nwg, ms * 1e3 / rounds, bad);
            }
    }
    // (5) a team of K workgroups on ONE XCD selected at run time (XCC_ID), flag-word barrier inside that XCD's L2
    {
        unsigned *d_flags, *d_ticket; CK(hipMalloc(&d_flags, 4096)); CK(hipMalloc(&d_ticket, 64));
        const char *tl[3] = {"plain loads", "nt loads", "16-B sc1 loads"};
        for (size_t words : {(size_t)0, total_words / 4, total_words})
            for (int K : {8, 32, 64, 128})
                for (int lf = 0; lf < 3; ++lf) {
                    if (words == 0 && lf) continue;
                    const int rounds = 2000, wpw = (int)(words / K);
                    float best = 1e9f; unsigned long long bad = 0; int err = 0;
                    for (int rep = 0; rep < 2; ++rep) {
                        CK(hipMemset(d_flags, 0, 4096)); CK(hipMemset(d_ticket, 0, 64)); CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_err, 0, 4));
                        CK(hipEventRecord(T.a, nullptr));
                        if (lf == 0) hipLaunchKernelGGL(team_kernel<0>, dim3(8 * K), dim3(256), 0, nullptr, 3, K, b0, b1, wpw, d_flags, d_ticket, 0u, rounds, d_err, d_bad);
                        else if (lf == 1) hipLaunchKernelGGL(team_kernel<1>, dim3(8 * K), dim3(256), 0, nullptr, 3, K, b0, b1, wpw, d_flags, d_ticket, 0u, rounds, d_err, d_bad);
                        else hipLaunchKernelGGL(team_kernel<2>, dim3(8 * K), dim3(256), 0, nullptr, 3, K, b0, b1, wpw, d_flags, d_ticket, 0u, rounds, d_err, d_bad);
                        CK(hipEventRecord(T.b, nullptr)); CK(hipDeviceSynchronize());
                        CK(hipEventElapsedTime(&ms, T.a, T.b)); best = ms < best ? ms : best;
                        CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
                    }
                    printf("team on one XCD (XCC_ID), %4zu KB, %-15s K = %3d: %.3f us per round, stale words %llu of %llu%s\n", words * 4 / 1024, words ? tl[lf] : "barrier only", K,
                           best * 1e3 / rounds, bad, (unsigned long long)words * rounds, err ? ", TIMEOUT" : "");
                }
    }
    return 0;
}
