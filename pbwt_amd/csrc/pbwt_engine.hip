// pbwt_engine.hip — host side of libpbwtgpu.so: the C ABI of include/pbwt_amd.h over the gfx950
// kernels in pbwt_kernels.h.  No CPU compute path lives here: every entry point fails loudly if
// there is no usable HIP device.
#include "../../include/pbwt_amd.h"
#include "pbwt_kernels.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <unordered_map>
#include <mutex>

using namespace pbwtk;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return 1;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
    } while (0)
#define CHK(expr) do { int _r = (expr); if (_r) return _r; } while (0)


// Environment switches.  The shipped library reads only those that select between bit-exact code paths the test-suite compares
// (PBWTAMD_SKEL, _SKEL_READ, _SKN, _SKN_MAXW, _PAIR1024, _NO_PACKED_FILL, _QS_BLOCKS, _THR_ROUNDS), the debugging aids (_GUARD, _POISON,
// _PROFILE, _TRACE_QS, _SHARD_TRACE) and PBWTAMD_LIB (Python binding).  Every tuning / probe switch (tile sizes, CU masks, stream
// priorities, occupancy pads, alternative kernels, the query sweep's pipeline knobs ...) goes through tune_env() and exists only in a
// measurement build (PBWTAMD_MEASURE_BUILD=1 -> -DPBWTAMD_MEASURE): DESIGN.md section 10 lists them.
#ifdef PBWTAMD_MEASURE
static inline const char *tune_env(const char *name) { return getenv(name); }
#else
static inline const char *tune_env(const char *) { return nullptr; }
#endif

// switches the test-suite toggles from one test to the next inside ONE process (bit-exact code paths, A/B in tests/test_gpu_parity.py): read at
// every call, not cached in a static (a cached switch silently keeps the first test's value for the rest of the run)
static inline int env_int(const char *name, int dflt) { const char *s = getenv(name); return s ? atoi(s) : dflt; }

// ------------------------------------------------------------------------------------ device memory
// PBWTAMD_GUARD=1 (debugging): every device buffer is mapped through the virtual-memory API so that it ENDS (to within its 256-byte
// alignment) at the end of its mapping with an unmapped granule behind it (=2: STARTS at the mapping's first byte, an unmapped
// granule before it): an access past a buffer faults at once instead of reading a neighbour.  Off: plain hipMalloc / hipFree.
struct GuardRec { void *va; size_t reserved; size_t mapped; void *mapAt; hipMemGenericAllocationHandle_t h; };
static std::unordered_map<void *, GuardRec> g_guard;
static std::mutex g_guard_mu;
static int guard_mode() { static const int v = getenv("PBWTAMD_GUARD") ? atoi(getenv("PBWTAMD_GUARD")) : 0; return v; }
static hipError_t dev_alloc(void **out, size_t n) {
    // PBWTAMD_POISON=<byte> (debugging): fresh buffers are filled with that byte instead of whatever the allocator hands out
    // (in practice zeros): a kernel that depends on memory it never wrote shows up in the parity tests
    static const int poison = getenv("PBWTAMD_POISON") ? atoi(getenv("PBWTAMD_POISON")) : -1;
    if (!guard_mode()) {
        hipError_t r0 = hipMalloc(out, n);
        if (r0 == hipSuccess && poison >= 0) { r0 = hipMemset(*out, poison, n); (void)hipDeviceSynchronize(); }   // (the null stream does not order with the engine's)
        return r0;
    }
    int dev = 0; hipError_t r = hipGetDevice(&dev); if (r != hipSuccess) return r;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; r = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum); if (r != hipSuccess) return r;
    n = std::max<size_t>(n, 1);
    GuardRec g; g.mapped = (n + gran - 1) / gran * gran; g.reserved = g.mapped + 2 * gran;
    r = hipMemAddressReserve(&g.va, g.reserved, gran, nullptr, 0); if (r != hipSuccess) return r;
    g.mapAt = (char *)g.va + gran;
    r = hipMemCreate(&g.h, g.mapped, &prop, 0); if (r != hipSuccess) return r;
    r = hipMemMap(g.mapAt, g.mapped, 0, g.h, 0); if (r != hipSuccess) return r;
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    r = hipMemSetAccess(g.mapAt, g.mapped, &ad, 1); if (r != hipSuccess) return r;
    const size_t back = (guard_mode() == 2) ? 0 : (g.mapped - (n + 255) / 256 * 256);
    *out = (char *)g.mapAt + back;
    if (poison >= 0) { r = hipMemset(g.mapAt, poison, g.mapped); (void)hipDeviceSynchronize(); if (r != hipSuccess) return r; }
    std::lock_guard<std::mutex> lk(g_guard_mu); g_guard[*out] = g;
    return hipSuccess;
}
// stream-ordered host-to-device copy.  hipMemcpy from pageable host memory INTO memory mapped through the virtual-memory API (guard
// mode) loses data on this image (tools/vmm_h2d_repro.hip: no pbwt code; every other 3 MB copy never arrives, whether or not the host waits
// for it; kernels and device-to-host copies on the same memory are fine) — so in guard mode the bytes travel through a hipMalloc bounce
// buffer and a copy kernel.  A debugging mode: the extra allocation and synchronisation do not matter there.
__global__ void guard_copy_kernel(unsigned char *dst, const unsigned char *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static hipError_t h2d_async(void *dst, const void *src, size_t n, hipStream_t st) {
    if (!guard_mode() || !n) return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
    void *bounce = nullptr;
    hipError_t r = hipMalloc(&bounce, n); if (r != hipSuccess) return r;
    r = hipMemcpyAsync(bounce, src, n, hipMemcpyHostToDevice, st);
    if (r == hipSuccess) { hipLaunchKernelGGL(guard_copy_kernel, dim3(256), dim3(256), 0, st, (unsigned char *)dst, (const unsigned char *)bounce, n); r = hipGetLastError(); }
    const hipError_t r2 = hipStreamSynchronize(st);
    (void)hipFree(bounce);
    return r != hipSuccess ? r : r2;
}
static hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    if (!guard_mode()) return hipFree(p);
    GuardRec g;
    { std::lock_guard<std::mutex> lk(g_guard_mu); auto it = g_guard.find(p); if (it == g_guard.end()) return hipFree(p); g = it->second; g_guard.erase(it); }
    (void)hipDeviceSynchronize();
    // The physical memory goes back; the ADDRESS RANGE stays reserved for the life of the process.  Measured on this image: with
    // hipMemAddressFree here, a later reservation can get the same addresses back and kernels then read stale data through the new
    // mapping (tests/test_gpu_parity.py read-side cases under PBWTAMD_GUARD=1: 6-9 of 13 wrong, a different set every run; 13 of 13 right,
    // run after run, once no guarded range is ever reused).  A debugging mode: 2^47 bytes of address space outlast any test run, and a
    // dangling pointer into a freed buffer now faults instead of hitting a recycled one.
    (void)hipMemUnmap(g.mapAt, g.mapped); (void)hipMemRelease(g.h);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------ engine
// small RAII holder for temporary device buffers
struct DevBufs {
    std::vector<void *> v;
    ~DevBufs() { for (void *p : v) if (p) (void)dev_free(p); }
    template <typename T> int alloc(T **out, size_t n) {
        void *p = nullptr;
        if (dev_alloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return fail("hipMalloc(%zu) failed", n * sizeof(T));
        v.push_back(p); *out = (T *)p; return 0;
    }
};

constexpr int QS_QUERIES_PER_WAVE = 1;     // queries a wave of the query sweep takes one after the other
constexpr int QS_SWEEP_CUS = 0;           // CUs the query sweep's stream is confined to (0: all) — see pbwtamd_match_sweep_sparse
// internal option: the caller reads the ring slots of the batch itself (forces the fill on the skeleton path)
constexpr unsigned OPT_INTERNAL_KEEP_STATES = 0x100u;
// internal option, with KEEP_STATES on a skeleton batch: the caller reads d of every slot but a[] only at the skeleton slots (0, 8, ...) — the fill
// neither reads nor writes a (half its bytes); the query sweep recovers the few ids it reports from the next skeleton state
constexpr unsigned OPT_INTERNAL_D_ONLY = 0x200u;

struct GraphKey { int with_d, sorted, ring, pair; hipGraphExec_t exec; };
struct Pending { bool valid = false; int ring = 0, kbase = 0, nb = 0; unsigned opts = 0; bool skel = false; bool sharded = false; bool early = false; int flushed = 0; /* leading sites whose consumers are enqueued already */ const uint32_t *cols = nullptr; /* the batch's bit columns */ };

// skeleton batches whose consumers need (d, y) of every site but not the haplotype ids (histogram sweep, pack3 through the
// sweep's bit columns): the fill writes d | y << 31 and no a — half the consumer stream's bytes (they are what slows the chain)
static inline bool packed_fill(const Pending &p) {
    static const bool no_fuse = tune_env("PBWTAMD_NO_YCOLS_FUSION") != nullptr;
    const bool off = getenv("PBWTAMD_NO_PACKED_FILL") != nullptr;
    // PBWTAMD_PACKED_CHECKSUM=1 (test aid): per-site checksums of d and y taken FROM the packed slots, so that the packed fill is checked at every position
    const bool packed_csum = env_int("PBWTAMD_PACKED_CHECKSUM", 0) != 0;
    const unsigned ids = (packed_csum ? 0u : PBWTAMD_OPT_CHECKSUM) | PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_LONG_RECS | 0x100u /* OPT_INTERNAL_KEEP_STATES */;
    return !off && !no_fuse && p.skel && (p.opts & PBWTAMD_OPT_WITHIN_HIST) && !(p.opts & ids);
}

// one rank of a position-sharded panel (pbwt_shard.inc)
struct ShardCtx {
    int rank = 0, world = 1, w0 = 0, Wl = 0;
    int tb[SHARD_MAX + 1] = {};                                 // tile boundaries
    int pb[SHARD_MAX + 1] = {};                               // position boundaries; pb[world] = M, unused entries INT_MAX
    ShardXch *xch = nullptr; bool xch_ext = false;            // own exchange block
    volatile int *h_err = nullptr;                            // pinned host mirror of the engine's device error word (refreshed behind every throttle event)
    ShardPeers peers = {};                                    // every rank's exchange block as mapped here (own = xch)
    int *peerA[SHARD_MAX] = {}, *peerD[SHARD_MAX] = {}; unsigned char *peerK[2][SHARD_MAX] = {};   // skeleton rings and key rows of every rank (own = this rank's)
    bool connected = false;
    int *SA = nullptr, *SD = nullptr; size_t nslot = 0;       // the skeleton ring: 2 rings of B/8+1 slots, a and d of the states 0, 8, 16, ... of a batch
    int2 *tbl = nullptr, *scan = nullptr; int *total = nullptr;              // chain scratch, rows indexed by global tile
    unsigned long long *agg = nullptr; unsigned *cnt = nullptr; unsigned cntEpoch = 0;
    unsigned e1 = 0, e2 = 0, e3 = 0;                          // epochs of f1 / f2 / f3
    int2 *ctbl = nullptr; unsigned long long *cagg = nullptr; unsigned *ccnt = nullptr; unsigned cEpoch = 0;   // consumer stream: hist rows + two-level scan state
    std::vector<long long> blkSite0, blkSites; unsigned long long *blkEnd = nullptr; size_t blkCap = 0;          // pack3 blocks this rank wrote
    bool full_state = true;                                   // slot 0 of the current ring is complete on this rank (pass start, after a replicated batch)
};

struct pbwtamd_engine {
    int device = 0, M = 0, Mpad = 0, wpc = 0, wpc64 = 0, W = 0, wpad = 0, E = 4, T = 1024, B = 0;
    hipStream_t stream = nullptr; bool own_stream = false;   // the launch chain
    hipStream_t s2 = nullptr;                                 // batch consumers
    hipEvent_t evChain[2] = {nullptr, nullptr}, evCons[2] = {nullptr, nullptr}; bool consRecorded[2] = {false, false}, chainRecorded[2] = {false, false};
    int2 *qs_bsum[2] = {nullptr, nullptr}; int qs_nblk = 0;   // query sweep: per ring, block summaries of every state of the batch (qs_blocksum_kernel), written by the batch's consumers
    int qs_bsum_sites[2] = {0, 0};          // ... and how many leading sites of the ring's batch they have summarised so far
    SkArgs *margs = nullptr, *margs_host = nullptr; size_t margs_cap = 0; int margs_half = 0; hipEvent_t evMargs[2] = {nullptr, nullptr};   // pbwtamd_pass_advance_many (panel 0 owns them)
    bool persist = false;                   // small panels (two-launch regime): all rounds of a batch in ONE launch (skel_persist_kernel) — set for the query cursor of the query sweep
    SkArgs *pargs = nullptr, *pargs_host = nullptr; unsigned *pbar = nullptr; unsigned pbar_epoch = 0; int pargs_half = 0; hipEvent_t evPargs[2] = {nullptr, nullptr};
    hipEvent_t evPreKeys = nullptr;         // read side: the next skeleton batch's rank directories and keys were derived ahead of time on another stream (query sweep); wait for this event instead
    int sub_rounds = 0; hipEvent_t evSub[8] = {}; long long evSub_n = 0;   // > 0: the consumers of a skeleton batch are enqueued every sub_rounds rounds, beside the rest of the batch's chain (query sweep)
    hipEvent_t evRounds[2] = {nullptr, nullptr}; bool roundsRecorded[2] = {false, false};   // everything the batch's consumers read is done (the last round's scatter into the OTHER ring may still wait for that ring's consumers)
    int ring = 0; Pending pend;
    int *A = nullptr, *D = nullptr; size_t strideA = 0, strideD = 0;      // 2 rings of B+1 slots
    int4 *summ = nullptr;
    int *ctl = nullptr;                     // [2]=device error flag
    Ctl *ctlblk = nullptr;                  // per-batch control block read by the step kernels
    long long *prof = nullptr;              // optional phase timestamps (PBWTAMD_PROFILE=1)
    int summ_cur = 0;                       // summary buffer holding the current site's tiles
    uint32_t *cols_stage = nullptr;         // (B+1) columns, for the host-buffer entry points
    unsigned long long *ycols = nullptr;    // (B+1) sorted bit columns (wpc64 words each)
    unsigned long long *colBytes = nullptr; // B+2
    P3Region *p3regs = nullptr;             // (B+2) x regions per column: the region-parallel pack3 encoder's per-region results
    unsigned long long *blockCount = nullptr; size_t blockCountCap = 0;
    unsigned long long *scal = nullptr;     // [0]=record total of the batch  [1]=yz bytes so far
    unsigned long long *hist = nullptr; int histlen = 0;
    unsigned long long *hist_rep = nullptr;  // HIST_REP replicas of the low histogram bins (streaming sweep), folded on read
    unsigned long long *csum = nullptr; int csum_sites = 0;               // 3 * csum_sites
    int4 *recs = nullptr; size_t recsCap = 0;
    uint8_t *yz = nullptr; size_t yzCap = 0;
    // pass state
    int k0 = 0, k_cur = 0, n_total = 0; bool prepared = false; bool pass_open = false;
    unsigned long long yz_bytes_host = 0; size_t yz_upper = 0;   // host-side upper bound of the packed bytes written
    // the true byte count travels back asynchronously (pinned ring + events): the bound is refreshed from the newest reading
    // that has landed, so the host never waits for the consumer stream unless the buffer really has to grow
    unsigned long long *h_used = nullptr; hipEvent_t evUsed[8] = {}; size_t usedWorstAfter[8] = {}; long long used_n = 0;
    std::vector<GraphKey> graphs; bool use_graph = true; bool lean = true; bool pair = true;
    bool pair1024 = false;
    bool skn = true;                        // skeleton rounds of two launches (hist, rank) when the panel has <= 128 tiles of 1024; PBWTAMD_SKN=0: K1/K2/K3
    bool ring_skel[2] = {false, false};     // the batch last advanced in each ring went through the skeleton path (its slots 1..7 mod 8 come from the fill)
    bool skel = true;                       // skeleton + fill (8 sites per round of K1/K2/K3 on the chain, the 7 states between filled beside it); PBWTAMD_SKEL=0: two-site chain
    uint32_t *xT = nullptr; size_t strideX = 0; int xTblocks = 0;   // transposed panel of the batch in flight (= xTr[ring])
    uint32_t *xTr[2] = {nullptr, nullptr}; // one per ring: the fill of batch n reads it while the chain transposes batch n+1
    int *skT = nullptr;                     // hist table of the round in flight, [W][256] {cnt, tail}
    unsigned long long *k2agg = nullptr; unsigned *k2cnt = nullptr; unsigned k2epoch = 0;   // two-level tile scan of wide panels (skel_k2_wide_kernel)
    unsigned char *keysR[2] = {nullptr, nullptr};         // per ring: the keys of states 0, 8, 16, ... of the batch ([B/8+1][Mpad]), kept for the fill
    unsigned *wflags = nullptr; size_t strideF = 0;           // fused fill + maxWithin: one bit per position and slot of a batch = "not decided in the fill" (sweep_resid_kernel clears what it reads)
    unsigned long long *nflag = nullptr, *h_nflag = nullptr; hipEvent_t evFlag = nullptr; bool flagPending = false;   // positions flagged (device total, pinned mirror)
    unsigned long long nflag_prev = 0; double flag_sites = 0; bool fuse_ok = true;      // ... a panel that leaves too many undecided goes back to the streaming sweep
    unsigned short *p16r = nullptr;                         // 2 rings of B+2 slots of the 16-bit hand-off (stride = strideD elements), allocated with the first batch that takes it
    int2 *fillGB[2] = {nullptr, nullptr};                   // per ring and round: [256] {G, base} per heap entry (skel_fillprep_kernel -> skel_fillseq_kernel)
    int2 *saveR[2] = {nullptr, nullptr}; size_t strideS = 0;  // per ring and round: scan[W][256] {before, carry}, total[256] (stride in int2)
    hipEvent_t tev[16] = {}; long long tev_n = 0; int thr_rounds = 28, thr_depth = 2;   // host throttle: an event every thr_rounds rounds, host at most thr_depth events ahead
    int *rankdirS = nullptr;                // read-side skeleton: zero-prefix directories of the batch's sorted columns [B+2][wpc64+1]
    bool keys_ready[2] = {false, false};     // slot-0 keys of the ring delivered by the previous batch's last round
    int q_lo = 0, q_hi = 0x7fffffff; bool q_part = false;   // query sweeps: only queries q_lo <= jj < q_hi (pbwtamd_set_query_range)
    bool prow = false; int W2 = 0;          // pair rows: skel_hist_kernel<4, true> + the scan on W2 = ceil(Wt / 2) rows
    int Wt = 0, skEPT = 4;                  // skeleton tiles: 256*skEPT positions, Wt of them; PBWTAMD_SKT=512|1024
    int skn_maxw = 48;                      // two-launch round (rank scans the tile table itself) up to this many tiles (measured: 5 k -28 %, 8 k -26 %, 10 k -9 %, 12 k -6 %, 16 k 0); PBWTAMD_SKN_MAXW
    bool summ_pair = false;                 // format of the current tile summaries (two-site keys or single site)
    uint32_t *zerocol = nullptr; long long sites_done = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t ev_used = 0; long long launches = 0;
    // record sink for pass_advance (host-buffer entry points)
    std::vector<pbwtamd_match> *rec_sink = nullptr; pbwtamd_report_fn rec_cb = nullptr;
    int longL = 0;                          // L of the -longWithin consumer (PBWTAMD_OPT_LONG_RECS)
    int *ystale = nullptr;                  // copy of the previous state's tagged a, for the k == N quirk of -longWithin
    std::vector<int32_t> nomatch_events;    // (jj, x, k[, isSparse]) of the last query sweep, in the reference's log order
    ShardCtx *sh = nullptr;          // position sharding across GPUs (pbwt_shard.inc): this engine is one rank of a panel
};

extern "C" int pbwtamd_abi_version(void) { return PBWTAMD_ABI_VERSION; }
#ifdef PBWTAMD_WALKSTAT
extern "C" int pbwtamd_measure_walkstat(unsigned long long *out, int reset) {      // measurement builds: the sweep's walk counters (pbwt_k_sweep.h: g_walkstat)
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(pbwtk::g_walkstat), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(pbwtk::g_walkstat), z, sizeof z)); }
    return 0;
}
#endif
extern "C" const char *pbwtamd_last_error(void) { return g_err.c_str(); }
extern "C" int pbwtamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void pbwtamd_free(void *p) { free(p); }
extern "C" int pbwtamd_engine_M(const pbwtamd_engine *e) { return e->M; }
extern "C" int pbwtamd_engine_wpc(const pbwtamd_engine *e) { return e->wpc; }
extern "C" int pbwtamd_engine_batch(const pbwtamd_engine *e) { return e->B; }

static int wpc_for(int M) { return ((M + 31) / 32 + 3) / 4 * 4; }
static inline int p3_regions(int M) { return ((M + 63) / 64 + 63) / 64; }   // regions of 64 words per column (region-parallel pack3 encoder)

static void shard_release(pbwtamd_engine *e);

extern "C" void pbwtamd_engine_destroy(pbwtamd_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->s2) (void)hipStreamSynchronize(e->s2);
    shard_release(e);
    if (e->s2) (void)hipStreamDestroy(e->s2);
    for (int i = 0; i < 16; ++i) if (e->tev[i]) (void)hipEventDestroy(e->tev[i]);
    for (int i = 0; i < 8; ++i) if (e->evUsed[i]) (void)hipEventDestroy(e->evUsed[i]);
    for (int i = 0; i < 8; ++i) if (e->evSub[i]) (void)hipEventDestroy(e->evSub[i]);
    for (int i = 0; i < 2; ++i) if (e->evPargs[i]) (void)hipEventDestroy(e->evPargs[i]);
    for (int i = 0; i < 2; ++i) if (e->evMargs[i]) (void)hipEventDestroy(e->evMargs[i]);
    if (e->margs_host) (void)hipHostFree(e->margs_host);
    if (e->margs) (void)dev_free(e->margs);
    if (e->pargs_host) (void)hipHostFree(e->pargs_host);
    if (e->pargs) (void)dev_free(e->pargs);
    if (e->pbar) (void)dev_free(e->pbar);
    if (e->h_used) (void)hipHostFree(e->h_used);
    if (e->h_nflag) (void)hipHostFree(e->h_nflag);
    if (e->evFlag) (void)hipEventDestroy(e->evFlag);
    for (int i = 0; i < 2; ++i) { if (e->evChain[i]) (void)hipEventDestroy(e->evChain[i]); if (e->evCons[i]) (void)hipEventDestroy(e->evCons[i]); if (e->evRounds[i]) (void)hipEventDestroy(e->evRounds[i]); }
    for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto &p : e->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    void *ptrs[] = {e->A, e->D, e->summ, e->ctl, (void *)e->ctlblk, (void *)e->prof, (void *)e->zerocol, (void *)e->ystale, (void *)e->xTr[0], (void *)e->xTr[1], (void *)e->keysR[0], (void *)e->keysR[1], (void *)e->saveR[0], (void *)e->saveR[1], (void *)e->fillGB[0], (void *)e->fillGB[1], (void *)e->p16r, (void *)e->wflags, (void *)e->nflag, (void *)e->rankdirS, (void *)e->skT, (void *)e->k2agg, (void *)e->k2cnt, e->cols_stage, e->ycols, e->colBytes, (void *)e->p3regs,
                    e->blockCount, e->scal, e->hist, e->hist_rep, e->csum, e->recs, e->yz};
    for (void *p : ptrs) if (p) (void)dev_free(p);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int pbwtamd_engine_create(pbwtamd_engine **out, int device, int M, int batch_sites, void *stream) {
    *out = nullptr;
    if (M < 1) return fail("pbwtamd_engine_create: M=%d", M);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("pbwtamd: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail("pbwtamd: device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    pbwtamd_engine *e = new pbwtamd_engine();
    e->device = device; e->M = M;
    // tile geometry: T = 256*E positions per workgroup, at most 1024 tiles
    // (latency-bound regime: the fewer positions per thread, the shorter the launch)
    e->E = 1;
    if (M > 262144) e->E = 4;                              // T = 1024 (two-site launches use 1024-thread workgroups)
    while (e->E < 16 && (M + BLOCK * e->E - 1) / (BLOCK * e->E) > 1024) e->E *= 2;
    if (const char *s = tune_env("PBWTAMD_E")) { int v = atoi(s); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) e->E = v; }
    if (const char *s = tune_env("PBWTAMD_T")) { int v = atoi(s); if (v == 256 || v == 1024 || v == 2048 || v == 4096) e->E = v / BLOCK; }
    e->T = BLOCK * e->E;
    e->W = (M + e->T - 1) / e->T;
    if (e->W > 1024) { delete e; return fail("pbwtamd: M=%d too large for this build (max %d)", M, 1024 * 4096); }   // nothing allocated yet
    e->Mpad = (M + 4095) / 4096 * 4096;                    // every tile geometry (256 / 1024 / 4096 positions) stays inside the padding
    e->wpad = (e->W + 63) / 64 * 64;
    e->wpc = wpc_for(M);
    e->wpc64 = e->wpc / 2;
    e->B = batch_sites > 0 ? batch_sites : 512;
    if (e->B & 1) ++e->B;                                  // two-site launches: even batches
    if (const char *s = tune_env("PBWTAMD_NO_GRAPH")) e->use_graph = !(atoi(s) != 0);
    if (const char *s = tune_env("PBWTAMD_LEAN")) e->lean = atoi(s) != 0;
    if (const char *s = tune_env("PBWTAMD_PAIR")) e->pair = atoi(s) != 0;
    if (const char *s = getenv("PBWTAMD_PAIR1024")) e->pair1024 = atoi(s) != 0;
    if (const char *s = getenv("PBWTAMD_SKEL")) e->skel = atoi(s) != 0;
    if (const char *s = getenv("PBWTAMD_SKN")) e->skn = atoi(s) != 0;
    if (const char *s = tune_env("PBWTAMD_PERSIST")) e->persist = atoi(s) != 0;
    // skeleton at every width the engine takes: up to 4096 tiles of 1024 positions (above 2048 tiles the two-level tile scan gives each of its
    // <= 64 co-resident workgroups 64 tiles instead of 32); PBWTAMD_SKEL_MAXM=<M> (A/B runs): the two-site chain above that width, as before round 3
    if (const char *sm = tune_env("PBWTAMD_SKEL_MAXM")) { if (M > atoi(sm)) e->skel = false; }
    int prLow = 0, prHigh = 0;
    (void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);  // numerically: low >= high
    if (const char *s = tune_env("PBWTAMD_NO_PRIO")) { if (atoi(s)) prLow = prHigh = 0; }
    if (stream) { e->stream = (hipStream_t)stream; e->own_stream = false; }
    else { if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prHigh) != hipSuccess) { delete e; return fail("hipStreamCreate failed"); } e->own_stream = true; }
    e->strideA = (size_t)e->Mpad;
    e->strideD = (size_t)e->Mpad + 64;
    const size_t slots = (size_t)e->B + 2;
#define ALLOC(ptr, bytes) do { hipError_t _e = dev_alloc((void **)&(ptr), (bytes)); if (_e != hipSuccess) { int r = fail("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(_e)); pbwtamd_engine_destroy(e); return r; } } while (0)
    ALLOC(e->A, 2 * slots * e->strideA * sizeof(int));
    ALLOC(e->D, 2 * slots * e->strideD * sizeof(int));
    ALLOC(e->summ, (size_t)3 * e->wpad * 3 * sizeof(int4));
    ALLOC(e->zerocol, (size_t)e->wpc * sizeof(uint32_t));
    ALLOC(e->ctl, 16 * sizeof(int));
    ALLOC(e->ctlblk, sizeof(Ctl));
#define ECHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int r = fail("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); pbwtamd_engine_destroy(e); return r; } } while (0)
    if (const char *s = getenv("PBWTAMD_PROFILE")) if (atoi(s)) { ALLOC(e->prof, (size_t)e->W * 8 * sizeof(long long)); ECHK(hipMemset(e->prof, 0, (size_t)e->W * 8 * sizeof(long long))); }
    ALLOC(e->cols_stage, 2 * (slots + 6) * e->wpc * sizeof(uint32_t));   // two halves of B+8 columns: batch + look-ahead, double-buffered by the host entry points
    ALLOC(e->ycols, slots * e->wpc64 * sizeof(unsigned long long));
    ALLOC(e->colBytes, (slots + 1) * sizeof(unsigned long long));
    ALLOC(e->p3regs, slots * (size_t)p3_regions(M) * sizeof(P3Region));
    ALLOC(e->scal, 8 * sizeof(unsigned long long));
    ALLOC(e->hist_rep, (size_t)HIST_REP * HIST_LBINS * sizeof(unsigned long long));
    if (e->skel) {
        // measured: smaller tiles = shorter per-workgroup latency chains, and every chain workgroup must fit into the hole a retiring
        // consumer workgroup leaves on a CU (LDS is allocated contiguously: a 41 KB rank workgroup of 1024 positions starves beside
        // 26 KB fill workgroups).  1 M, T = 512 vs 1024: 6.25 vs 6.84 us/site; 500 k: 3.99 vs 4.32
        e->skEPT = (M <= 56000) ? 1 : 2;                       // 256- against 512-position tiles, end to end: 50 k 1.33 vs 1.37 us/site, 70 k 1.52 vs 1.45
        // 8 193 .. 12 288 haplotypes: the two-launch round (the rank launch scans the tile table itself) on 17-24 tiles of 512 positions instead of
        // 33-48 of 256 — half the rows in front of every rank workgroup: 1.30 -> 1.21 us/site at 10 k, 1.31 -> 1.22 at 12 k; equal at 5-8 k,
        // worse at 2 k (0.98 -> 1.05) and from 13 k on, where three launches on 256-position tiles take over (1.20)
        if (M > 8192 && M <= 12288) e->skEPT = 2;
        if (const char *sv = tune_env("PBWTAMD_SKT")) e->skEPT = (atoi(sv) == 256) ? 1 : (atoi(sv) == 512) ? 2 : 4;

        // pair rows carry 512-position tiles up to 4096 rows of pairs = 2^22 haplotypes, every width the skeleton takes (the wide scan: <= 64
        // workgroups of 32 rows up to 2048 rows, of 64 above): against 1024-position tiles, end to end 7.42 -> 6.61 us/site at 1.1 M, 9.77 -> 8.90
        // at 1.5 M, 12.68 -> 11.60 at 2.0 M, 13.90 -> 12.57 at 2.2 M, 18.74 -> 17.00 at 3 M, 24.68 -> 22.40 at 4 M.  1024-position tiles are left
        // to measurement builds (PBWTAMD_SKT=1024 / PBWTAMD_PROW_MAX)
        static const int prow_max = tune_env("PBWTAMD_PROW_MAX") ? std::min(4096, atoi(tune_env("PBWTAMD_PROW_MAX"))) : 4096;
        const bool pairs_reach = e->skEPT == 2 && (M + 1023) / 1024 <= prow_max;
        if (M > 256 * e->skEPT * 2048 && !pairs_reach) e->skEPT = 4;   // skel_k2_kernel scans at most 2048 tiles per key
        if (const char *sv = getenv("PBWTAMD_SKN_MAXW")) e->skn_maxw = std::min(atoi(sv), SKN_MAXW);
        e->Wt = (M + 256 * e->skEPT - 1) / (256 * e->skEPT);
        e->strideX = (size_t)e->Mpad; e->xTblocks = (e->B + 8 + 31) / 32 + 1;
        ALLOC(e->xTr[0], (size_t)e->xTblocks * e->strideX * sizeof(uint32_t));
        ALLOC(e->xTr[1], (size_t)e->xTblocks * e->strideX * sizeof(uint32_t));
        if (const char *sv = getenv("PBWTAMD_THR_ROUNDS")) e->thr_rounds = atoi(sv);
        if (const char *sv = tune_env("PBWTAMD_THR_DEPTH")) e->thr_depth = std::max(1, std::min(atoi(sv), 15));
        for (int i = 0; i < 16; ++i) ECHK(hipEventCreateWithFlags(&e->tev[i], hipEventDisableTiming));
        {
            const int rounds = e->B / 8 + 1;
            // pair rows (512-position tiles): the scan over the tiles runs on 1024-position pairs — half the rows — and the rank / fill
            // workgroup of an odd tile folds the first tile's row in.  Rows 512 < W2 <= 1024: the wide scan, <= 32 workgroups; 136 < W2 <= 512
            // (139 k < M <= 524 k): the one-level scan on half the rows — measured -2 % at 140 k, -3 % at 160-250 k, -12 % at 300 k, -8 % at
            // 400 k, -6 % at 500 k end to end; +4..6 % at 100-120 k, where the hist workgroup of two tiles costs more than the shorter scan saves
            static const bool pair_rows = !(tune_env("PBWTAMD_PAIR_ROWS") && !atoi(tune_env("PBWTAMD_PAIR_ROWS")));
            e->W2 = (e->Wt + 1) / 2;
            static const int prow_min = tune_env("PBWTAMD_PROW_MIN") ? atoi(tune_env("PBWTAMD_PROW_MIN")) : 136;
            static const bool prow_ept1 = tune_env("PBWTAMD_PROW_EPT1") && atoi(tune_env("PBWTAMD_PROW_EPT1"));   // measurement builds: pairs of 256-position tiles
            e->prow = pair_rows && (e->skEPT == 2 || (e->skEPT == 1 && prow_ept1)) && e->W2 > prow_min && e->W2 <= prow_max;
            if (e->skEPT == 2 && e->Wt > 2048 && !e->prow) { const int r = fail("pbwtamd_engine_create: %d tiles of 512 positions need pair rows", e->Wt); pbwtamd_engine_destroy(e); return r; }
            e->strideS = e->prow ? (size_t)SKK * e->W2 * 2 + SKK / 2 : (size_t)SKK * e->Wt + SKK / 2;
            for (int i = 0; i < 2; ++i) {
                ALLOC(e->keysR[i], (size_t)(rounds + 1) * e->Mpad);
                ALLOC(e->saveR[i], (size_t)rounds * e->strideS * sizeof(int2));
                ALLOC(e->fillGB[i], (size_t)rounds * SKK * sizeof(int2));
            }
            // the 16-bit hand-off ring (run_consumers): 2 x (B + 2) slots of strideD 16-bit words — here rather than with the first batch that takes it
            // (a 0.2-2 GB hipMalloc inside a pass showed up as a 10 % outlier in one run out of four at 100 k haplotypes)
            if (e->skEPT <= 2 && env_int("PBWTAMD_P16", 1) != 0) ALLOC(e->p16r, (size_t)2 * (e->B + 2) * e->strideD * sizeof(unsigned short));
        }
        ALLOC(e->skT, (size_t)(e->Wt + 1) * SKK * sizeof(int2));
        ALLOC(e->k2agg, (size_t)64 * SKK * sizeof(unsigned long long));
        ALLOC(e->k2cnt, 64);
        ECHK(hipMemsetAsync(e->k2cnt, 0, 64, e->stream));
        ALLOC(e->rankdirS, (size_t)(e->B + 2) * (e->wpc64 + 1) * sizeof(int));
    }
#undef ALLOC
    ECHK(hipMemsetAsync(e->A, 0, 2 * slots * e->strideA * sizeof(int), e->stream));
    ECHK(hipMemsetAsync(e->D, 0, 2 * slots * e->strideD * sizeof(int), e->stream));
    {   // consumers yield to the dependent chain: low priority; and while the chain is the bottleneck (narrow panels) they run on
        // the first 5/8 of the CUs only, so the chain's workgroups find whole shader arrays without scattered-store traffic in
        // their memory pipelines (measured +3 % at M = 100 k and +13 % at 250 k with 160 of 256 CUs; at M = 1 M the consumers are the bottleneck: no mask)
        int ncu_dev = 0; (void)hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, device);
        // re-measured with the register-light tile scan (us/site, none / 160 / 192 CUs): 150 k 2.13 / 1.88 / 1.91, 200 k 2.40 / 2.11 / 2.15,
        // 250 k 2.69 / 2.35 / 2.40, 300 k 2.99 / 2.97 / 2.97, 400 k 3.43 / 3.39 / 3.35, 500 k 3.83 / . / 3.78, 700 k 4.66 / . / 4.66, 1 M 5.84 / . / 6.56
        int ncus = (ncu_dev < 64 || ncu_dev > 256) ? 0 : (M <= 270000) ? ncu_dev * 5 / 8 : (M <= 600000) ? ncu_dev * 3 / 4 : 0;
        if (const char *s = tune_env("PBWTAMD_S2_CUS")) ncus = atoi(s);
        if (ncus > 0) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < std::min(ncus, 256); ++i) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&e->s2, 8, mask) != hipSuccess) { (void)hipGetLastError(); e->s2 = nullptr; }
        }
        if (!e->s2) ECHK(hipStreamCreateWithPriority(&e->s2, hipStreamNonBlocking, prLow));
    }
    for (int i = 0; i < 2; ++i) { ECHK(hipEventCreateWithFlags(&e->evChain[i], hipEventDisableTiming)); ECHK(hipEventCreateWithFlags(&e->evCons[i], hipEventDisableTiming)); ECHK(hipEventCreateWithFlags(&e->evRounds[i], hipEventDisableTiming)); }
    ECHK(hipMemsetAsync(e->ctl, 0, 16 * sizeof(int), e->stream));
    ECHK(hipMemsetAsync(e->scal, 0, 8 * sizeof(unsigned long long), e->stream));
    ECHK(hipMemsetAsync(e->zerocol, 0, (size_t)e->wpc * sizeof(uint32_t), e->stream));
    ECHK(hipStreamSynchronize(e->stream));
#undef ECHK
    *out = e;
    return 0;
}

static int flush_pending(pbwtamd_engine *e);

extern "C" int pbwtamd_sync(pbwtamd_engine *e) {
    HIPCHK(hipSetDevice(e->device));
    CHK(flush_pending(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->s2));
    int err = 0;
    HIPCHK(hipMemcpy(&err, e->ctl + 2, sizeof(int), hipMemcpyDeviceToHost));
    if (err) return fail("pbwtamd: device-side error flag %d (1=histogram range, 2/3=malformed packed column, 4=yz buffer overflow, 5=tile scan of a wide panel timed out waiting for its workgroups, 6/7=a position-sharded rank timed out waiting for its peers, 9=a peer reported its own failure)", err);
    return 0;
}

// ------------------------------------------------------------------------------------ small kernels
// start of a batch: publish the control block and rotate the tile summaries so that the current
// site's summaries sit in buffer 0 (step j reads buffer j%3), with buffer 1 cleared for accumulation
// (`n` = int4 entries per summary buffer: wpad for single-site steps, 3*wpad for two-site steps)
__global__ __launch_bounds__(256) void set_ctl_kernel(Ctl *ctl, int kbase, int n_total, const uint32_t *cols, const uint32_t *zerocol, int4 *summ, int n, int cur) {
    if (threadIdx.x == 0) { ctl->kbase = kbase; ctl->n_total = n_total; ctl->cols = cols; ctl->zerocol = zerocol; ctl->pad0 = 0; ctl->pad1 = 0; }
    if (cur != 0) {
        for (int i = threadIdx.x; i < n; i += 256) summ[i] = summ[(size_t)cur * n + i];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 256) summ[(size_t)n + i] = make_int4(0, 0, 0, 0);
}
__global__ void add_base_kernel(unsigned long long *v, size_t n, const unsigned long long *base) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += *base;
}
__global__ void bump_kernel(unsigned long long *acc, const unsigned long long *add, unsigned long long cap, int *err) {
    *acc += *add;
    if (*acc > cap) atomicExch(err, 4);
}

static inline int *ringA(pbwtamd_engine *e, int r) { return e->A + (size_t)r * ((size_t)e->B + 2) * e->strideA; }
static inline int *ringD(pbwtamd_engine *e, int r) { return e->D + (size_t)r * ((size_t)e->B + 2) * e->strideD; }

template <int E, bool WITH_D, bool SORTED>
static void launch_step(pbwtamd_engine *e, int ring, int j) {
    StepArgs g;
    int *A = ringA(e, ring), *D = ringD(e, ring);
    g.a_in = A + (size_t)j * e->strideA;        g.d_in = D + (size_t)j * e->strideD;
    g.a_out = A + (size_t)(j + 1) * e->strideA; g.d_out = D + (size_t)(j + 1) * e->strideD;
    g.ctl = e->ctlblk; g.summ = e->summ; g.prof = e->prof; g.wpc = e->wpc;
    g.j = j; g.M = e->M; g.W = e->W; g.wpad = e->wpad;
    if (E == 1 && e->lean) {
        if (e->W <= 256) hipLaunchKernelGGL((step1_kernel<WITH_D, SORTED, 1>), dim3(e->W), dim3(BLOCK), 0, e->stream, g);
        else if (e->W <= 512) hipLaunchKernelGGL((step1_kernel<WITH_D, SORTED, 2>), dim3(e->W), dim3(BLOCK), 0, e->stream, g);
        else hipLaunchKernelGGL((step1_kernel<WITH_D, SORTED, 4>), dim3(e->W), dim3(BLOCK), 0, e->stream, g);
    }
    else hipLaunchKernelGGL((step_kernel<E, WITH_D, SORTED>), dim3(e->W), dim3(BLOCK), 0, e->stream, g);
}

static void launch_step_dyn(pbwtamd_engine *e, int ring, int j, bool with_d, bool sorted) {
#define CASE(EE)                                                                   \
    if (e->E == EE) {                                                              \
        if (with_d && sorted) launch_step<EE, true, true>(e, ring, j);             \
        else if (with_d) launch_step<EE, true, false>(e, ring, j);                 \
        else if (sorted) launch_step<EE, false, true>(e, ring, j);                 \
        else launch_step<EE, false, false>(e, ring, j);                            \
        return;                                                                    \
    }
    CASE(1) CASE(2) CASE(4) CASE(8) CASE(16)
#undef CASE
}

static void launch_step2(pbwtamd_engine *e, int ring, int jl, bool with_d) {
    Step2Args g;
    int *A = ringA(e, ring), *D = ringD(e, ring);
    g.a_in = A + (size_t)(2 * jl) * e->strideA;      g.d_in = D + (size_t)(2 * jl) * e->strideD;
    g.a_mid = A + (size_t)(2 * jl + 1) * e->strideA; g.d_mid = D + (size_t)(2 * jl + 1) * e->strideD;
    g.a_out = A + (size_t)(2 * jl + 2) * e->strideA; g.d_out = D + (size_t)(2 * jl + 2) * e->strideD;
    g.ctl = e->ctlblk; g.summ = e->summ; g.prof = e->prof; g.wpc = e->wpc; g.jl = jl; g.M = e->M; g.W = e->W; g.wpad = e->wpad;
#define L2(WD, SP, NT, EE) hipLaunchKernelGGL((step2_kernel<WD, SP, NT, EE>), dim3(e->W), dim3(NT), 0, e->stream, g)
    if (e->T == 1024 && e->pair1024) { if (with_d) L2(true, 1, 1024, 1); else L2(false, 1, 1024, 1); }     // 16-wave workgroups (opt-in)
    else if (e->T == 4096) { if (with_d) L2(true, 1, 1024, 4); else L2(false, 1, 1024, 4); }              // one 16-wave workgroup per CU, 4 positions per thread
    else if (e->T == 2048) {                               // 8-wave workgroups, 4 positions per thread; SPT summaries per thread cover W <= 512 * SPT tiles
        if (with_d) { if (e->W <= 512) L2(true, 1, 512, 4); else L2(true, 2, 512, 4); }
        else        { if (e->W <= 512) L2(false, 1, 512, 4); else L2(false, 2, 512, 4); }
    }
    else if (e->T == 1024) {                               // 4 positions per thread, 4-wave workgroups
        if (with_d) { if (e->W <= 256) L2(true, 1, 256, 4); else if (e->W <= 512) L2(true, 2, 256, 4); else L2(true, 4, 256, 4); }
        else        { if (e->W <= 256) L2(false, 1, 256, 4); else if (e->W <= 512) L2(false, 2, 256, 4); else L2(false, 4, 256, 4); }
    }
    else if (with_d) { if (e->W <= 256) L2(true, 1, 256, 1); else if (e->W <= 512) L2(true, 2, 256, 1); else L2(true, 4, 256, 1); }
    else             { if (e->W <= 256) L2(false, 1, 256, 1); else if (e->W <= 512) L2(false, 2, 256, 1); else L2(false, 4, 256, 1); }
#undef L2
}

static int get_graph(pbwtamd_engine *e, bool with_d, bool sorted, int ring, bool pair, hipGraphExec_t *out) {
    for (auto &g : e->graphs) if (g.with_d == (int)with_d && g.sorted == (int)sorted && g.ring == ring && g.pair == (int)pair) { *out = g.exec; return 0; }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    if (pair) { for (int jl = 0; jl < e->B / 2; ++jl) launch_step2(e, ring, jl, with_d); }
    else { for (int j = 0; j < e->B; ++j) launch_step_dyn(e, ring, j, with_d, sorted); }
    const hipError_t ec = hipStreamEndCapture(e->stream, &graph);      // always leave capture mode
    if (ec != hipSuccess || !graph) return fail("hipStreamEndCapture: %s", hipGetErrorString(ec));
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) return fail("hipGraphInstantiate: %s", hipGetErrorString(ei));
    e->graphs.push_back(GraphKey{(int)with_d, (int)sorted, ring, (int)pair, exec});
    *out = exec;
    return 0;
}

// ------------------------------------------------------------------------------------ pass
extern "C" int pbwtamd_pass_begin(pbwtamd_engine *e, const int32_t *aInit, int k0, int n_total) {
    HIPCHK(hipSetDevice(e->device));
    if (n_total < k0) return fail("pbwtamd_pass_begin: n_total %d < k0 %d", n_total, k0);
    if (aInit) {                                           // it drives device gathers and scatters: must be a permutation of [0, M)
        std::vector<bool> seen((size_t)e->M, false);
        for (int i = 0; i < e->M; ++i) {
            const int v = aInit[i];
            if (v < 0 || v >= e->M || seen[(size_t)v]) return fail("pbwtamd_pass_begin: the start order is not a permutation of [0, %d) (entry %d = %d)", e->M, i, v);
            seen[(size_t)v] = true;
        }
    }
    e->pend.valid = false;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->s2));
    if (e->k2cnt) { HIPCHK(hipMemsetAsync(e->k2cnt, 0, 64, e->stream)); e->k2epoch = 0; }   // arrival counter and host epoch restart together
    if (e->pbar) { HIPCHK(hipMemsetAsync(e->pbar, 0, 64, e->stream)); e->pbar_epoch = 0; }
    if (e->sh) { e->sh->full_state = true; e->sh->blkSite0.clear(); e->sh->blkSites.clear(); }
    e->k0 = k0; e->k_cur = k0; e->n_total = n_total; e->prepared = false; e->pass_open = true;
    e->ring = 0; e->consRecorded[0] = e->consRecorded[1] = false; e->chainRecorded[0] = e->chainRecorded[1] = false; e->roundsRecorded[0] = e->roundsRecorded[1] = false;
    e->keys_ready[0] = e->keys_ready[1] = false;
    if (aInit) HIPCHK(h2d_async(e->A, aInit, sizeof(int) * (size_t)e->M, e->stream));
    const int nb = (e->Mpad + 1 + 255) / 256;
    hipLaunchKernelGGL(init_state_kernel, dim3(nb), dim3(256), 0, e->stream, e->A, e->D, e->M, e->Mpad, k0, aInit ? 0 : 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(e->ctl, 0, 16 * sizeof(int), e->stream));
    HIPCHK(hipMemsetAsync(e->scal, 0, 8 * sizeof(unsigned long long), e->stream));
    const int nsites = n_total - k0 + 1;
    if (nsites > e->csum_sites) {
        if (e->csum) HIPCHK(dev_free(e->csum));
        HIPCHK(dev_alloc((void **)&e->csum, (size_t)3 * nsites * sizeof(unsigned long long)));
        e->csum_sites = nsites;
    }
    HIPCHK(hipMemsetAsync(e->csum, 0, (size_t)3 * e->csum_sites * sizeof(unsigned long long), e->stream));
    const int hl = n_total + 2;
    if (hl > e->histlen) {
        if (e->hist) HIPCHK(dev_free(e->hist));
        HIPCHK(dev_alloc((void **)&e->hist, (size_t)hl * sizeof(unsigned long long)));
        e->histlen = hl;
    }
    HIPCHK(hipMemsetAsync(e->hist, 0, (size_t)e->histlen * sizeof(unsigned long long), e->stream));
    HIPCHK(hipMemsetAsync(e->hist_rep, 0, (size_t)HIST_REP * HIST_LBINS * sizeof(unsigned long long), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->yz_bytes_host = 0; e->yz_upper = 0; e->used_n = 0;
    e->ev_used = 0; e->launches = 0; e->sites_done = 0;
    return 0;
}

static int ensure_blockcount(pbwtamd_engine *e, size_t n) {
    if (n <= e->blockCountCap) return 0;
    if (e->blockCount) HIPCHK(dev_free(e->blockCount));
    HIPCHK(dev_alloc((void **)&e->blockCount, n * sizeof(unsigned long long)));
    e->blockCountCap = n;
    return 0;
}

// maxWithin sweep over `nsites` slots of (A, D) (sites kbase..) on stream st: histogram or records
static int run_within(pbwtamd_engine *e, hipStream_t st, const int *A, const int *D, int kbase, int nsites, int final_site, unsigned opts, bool packed = false, bool ycin = false, const unsigned short *p16 = nullptr, int p16_clip = P16_ESC) {
    SweepArgs g;
    g.A = A; g.D = D; g.strideA = e->strideA; g.strideD = e->strideD;
    g.M = e->M; g.kbase = kbase; g.final_site = final_site;
    g.blockCount = nullptr; g.recs = nullptr; g.hist = e->hist; g.histlen = e->histlen; g.err = e->ctl + 2;
    g.P16 = p16; g.stride16 = e->strideD; g.clip = p16_clip;
    static const bool no_fuse = tune_env("PBWTAMD_NO_YCOLS_FUSION") != nullptr;
    g.ycols = (!no_fuse && final_site < 0 && (opts & PBWTAMD_OPT_PACK3) && (opts & PBWTAMD_OPT_WITHIN_HIST)) ? e->ycols : nullptr; g.wpc64 = e->wpc64;
#ifdef PBWTAMD_MEASURE
    static const int sweep_dbg = getenv("PBWTAMD_DEBUG_SWEEP") ? atoi(getenv("PBWTAMD_DEBUG_SWEEP")) : 0; g.dbg = sweep_dbg;
#endif
    const int tiles = (e->M + BLOCK - 1) / BLOCK;
    dim3 grid(tiles, nsites);
    static const int sweep_it = tune_env("PBWTAMD_SWEEP_ITERS") ? atoi(tune_env("PBWTAMD_SWEEP_ITERS")) : 0;
    g.nvb = tiles;
    const int iters = sweep_it > 0 ? sweep_it : (tiles >= 64 ? 4 : 1);
    static const bool old_sweep = tune_env("PBWTAMD_OLD_SWEEP") != nullptr;   // the walking form of the histogram sweep (A/B runs)
    if ((opts & PBWTAMD_OPT_WITHIN_HIST) && !old_sweep) {  // streaming form: a wave per 256 positions
        const int ngroups = (e->M / 256 + 1 + WAVES - 1) / WAVES;    // 1024-position groups (a wave per 256 positions)
        g.hist_rep = e->hist_rep; g.iters = ngroups >= 64 ? 8 : (ngroups >= 8 ? 2 : 1);
        dim3 gs((ngroups + g.iters - 1) / g.iters, nsites);
#ifdef PBWTAMD_MEASURE
        if (packed && ycin) {                               // the fill emitted the allele columns: the sweep reads them first and loads d | y only where it has to
            g.ycols = e->ycols;
            const int nw = (e->M + 63) / 64;
            hipLaunchKernelGGL((sweep_hist_kernel<true, true>), dim3((nw + 64 * WAVES - 1) / (64 * WAVES), nsites), dim3(BLOCK), 0, st, g);
        } else
#endif
        if (packed && p16) hipLaunchKernelGGL((sweep_hist_kernel<true, false, true>), gs, dim3(BLOCK), 0, st, g);
        else if (packed) hipLaunchKernelGGL((sweep_hist_kernel<true>), gs, dim3(BLOCK), 0, st, g);
        else hipLaunchKernelGGL((sweep_hist_kernel<false>), gs, dim3(BLOCK), 0, st, g);
        HIPCHK(hipGetLastError());
    }
#ifdef PBWTAMD_MEASURE
    else if (opts & PBWTAMD_OPT_WITHIN_HIST) {
        dim3 gh((tiles + iters - 1) / iters, nsites);
#define SWEEP_HIST(P, I) hipLaunchKernelGGL((sweep_within_kernel<2, P, I>), gh, dim3(BLOCK), 0, st, g)
        if (packed) { if (iters == 8) SWEEP_HIST(true, 8); else if (iters == 4) SWEEP_HIST(true, 4); else if (iters == 2) SWEEP_HIST(true, 2); else { gh.x = tiles; SWEEP_HIST(true, 1); } }
        else { if (iters == 8) SWEEP_HIST(false, 8); else if (iters == 4) SWEEP_HIST(false, 4); else if (iters == 2) SWEEP_HIST(false, 2); else { gh.x = tiles; SWEEP_HIST(false, 1); } }
#undef SWEEP_HIST
        HIPCHK(hipGetLastError());
    }
#else
    (void)iters; (void)old_sweep;
#endif
    if (opts & PBWTAMD_OPT_WITHIN_RECS) {
        const size_t nblk = (size_t)tiles * nsites;
        CHK(ensure_blockcount(e, nblk));
        g.blockCount = e->blockCount;
        hipLaunchKernelGGL((sweep_within_kernel<0>), grid, dim3(BLOCK), 0, st, g);
        hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, e->blockCount, nblk, e->scal, 0ULL);
        HIPCHK(hipGetLastError());
        unsigned long long total = 0;
        HIPCHK(hipMemcpyAsync(&total, e->scal, sizeof total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (total > e->recsCap) {
            if (e->recs) HIPCHK(dev_free(e->recs));
            e->recsCap = (size_t)(total + total / 4 + 1024);
            HIPCHK(dev_alloc((void **)&e->recs, e->recsCap * sizeof(int4)));
        }
        if (total) {
            g.recs = e->recs;
            hipLaunchKernelGGL((sweep_within_kernel<1>), grid, dim3(BLOCK), 0, st, g);
            HIPCHK(hipGetLastError());
            std::vector<pbwtamd_match> tmp;
            std::vector<pbwtamd_match> *dst = e->rec_sink ? e->rec_sink : &tmp;
            const size_t old = dst->size();
            dst->resize(old + total);
            HIPCHK(hipMemcpyAsync(dst->data() + old, e->recs, total * sizeof(int4), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (e->rec_cb) {
                for (size_t r = old; r < old + total; ++r) { const pbwtamd_match &m = (*dst)[r]; e->rec_cb(m.ai, m.bi, m.start, m.end); }
                if (dst == e->rec_sink) dst->resize(old);       // callback mode keeps nothing
            }
        }
    }
    return 0;
}

// -longWithin L over `nsites` slots (records only)
static int run_long(pbwtamd_engine *e, hipStream_t st, const int *A, const int *D, const int *Ystale, int kbase, int nsites, int final_site) {
    LongArgs g;
    g.A = A; g.D = D; g.strideA = e->strideA; g.strideD = e->strideD; g.Ystale = Ystale ? Ystale : A;
    g.M = e->M; g.kbase = kbase; g.final_site = final_site; g.L = e->longL;
    const int tiles = (e->M + BLOCK - 1) / BLOCK;
    dim3 grid(tiles, nsites);
    const size_t nblk = (size_t)tiles * nsites;
    CHK(ensure_blockcount(e, nblk));
    g.blockCount = e->blockCount; g.recs = nullptr;
    hipLaunchKernelGGL((sweep_long_kernel<0>), grid, dim3(BLOCK), 0, st, g);
    hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, e->blockCount, nblk, e->scal, 0ULL);
    HIPCHK(hipGetLastError());
    unsigned long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, e->scal, sizeof total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (total > e->recsCap) {
        if (e->recs) HIPCHK(dev_free(e->recs));
        e->recsCap = (size_t)(total + total / 4 + 1024);
        HIPCHK(dev_alloc((void **)&e->recs, e->recsCap * sizeof(int4)));
    }
    if (total) {
        g.recs = e->recs;
        hipLaunchKernelGGL((sweep_long_kernel<1>), grid, dim3(BLOCK), 0, st, g);
        HIPCHK(hipGetLastError());
        std::vector<pbwtamd_match> tmp;
        std::vector<pbwtamd_match> *dst = e->rec_sink ? e->rec_sink : &tmp;
        const size_t old = dst->size();
        dst->resize(old + total);
        HIPCHK(hipMemcpyAsync(dst->data() + old, e->recs, total * sizeof(int4), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (e->rec_cb) {
            for (size_t r = old; r < old + total; ++r) { const pbwtamd_match &m = (*dst)[r]; e->rec_cb(m.ai, m.bi, m.start, m.end); }
            if (dst == e->rec_sink) dst->resize(old);
        }
    }
    return 0;
}

static int ensure_yz(pbwtamd_engine *e, hipStream_t st, size_t cap) {
    if (cap <= e->yzCap) return 0;
    uint8_t *n = nullptr;
    HIPCHK(dev_alloc((void **)&n, cap));
    if (e->yz) {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(n, e->yz, e->yzCap, hipMemcpyDeviceToDevice));
        HIPCHK(dev_free(e->yz));
    }
    e->yz = n; e->yzCap = cap;
    return 0;
}

// pack3-encode the y columns (tags) of `nsites` slots of A and append to the engine's yz buffer.
// No host sync in the steady state: the host tracks an upper bound of the bytes used (worst case one
// byte per position) and only reads the true count back when that bound would exceed the capacity.
__global__ void pack3_offsets_kernel(unsigned long long *colBytes, size_t n, const unsigned long long *base, unsigned long long *batchTotal,
                                     unsigned long long *acc, unsigned long long cap, int *err) {
    // colBytes holds exclusive offsets inside the batch (scan done); rebase them and bump the running total
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long b = *base;
    if (i < n) colBytes[i] += b;
    if (i == 0) { if (b + *batchTotal > cap) atomicExch(err, 4); }
    (void)acc;
}
// pack3v2_kernel<MODE, NT, IT>: NT/64 waves of 64*IT words each cover the column
// region-parallel encoder (p3r_*): sizes of nsites columns into colBytes, then (after the caller's scan over the columns) the bytes
static void launch_p3r_sizes(hipStream_t st, int nsites, const unsigned long long *ycols, int wpc64, int M, P3Region *regs, unsigned long long *colBytes) {
    const int R = p3_regions(M);
    hipLaunchKernelGGL((p3r_scan_kernel<1>), dim3((R + WAVES - 1) / WAVES, nsites), dim3(BLOCK), 0, st, ycols, wpc64, M, R, regs);
    hipLaunchKernelGGL(p3r_combine_kernel, dim3((nsites + WAVES - 1) / WAVES), dim3(BLOCK), 0, st, M, R, 64, nsites, regs, colBytes);
}
static void launch_p3r_emit(hipStream_t st, int nsites, const unsigned long long *ycols, int wpc64, int M, const P3Region *regs, const unsigned long long *colOff, uint8_t *out) {
    const int R = p3_regions(M);
    hipLaunchKernelGGL((p3r_emit_kernel<1>), dim3((R + WAVES - 1) / WAVES, nsites), dim3(BLOCK), 0, st, ycols, wpc64, M, R, regs, colOff, out);
}

#ifdef PBWTAMD_MEASURE
template <int MODE>
static void launch_pack3v2(hipStream_t st, int nsites, const unsigned long long *ycols, int wpc64, int M, unsigned long long *colBytes, uint8_t *out) {
    const int nw = (M + 63) / 64;
#define P3(NT, IT) hipLaunchKernelGGL((pack3v2_kernel<MODE, NT, IT>), dim3(nsites), dim3(NT), 0, st, ycols, wpc64, M, colBytes, out)
    // as many waves and as few 64-word iterations per wave as the column allows: a wave's iterations are a serial instruction
    // stream with nothing to hide its latency behind (measured at M = 100 k: 4 waves x 8 iterations 32 + 91 us per batch)
    if (nw <= 256) P3(256, 1); else if (nw <= 512) P3(256, 2); else if (nw <= 1024) P3(1024, 1); else if (nw <= 2048) P3(1024, 2);
    else if (nw <= 4096) P3(1024, 4); else if (nw <= 8192) P3(1024, 8); else if (nw <= 16384) P3(1024, 16); else P3(1024, 64);
#undef P3
}
#endif

static int run_pack3(pbwtamd_engine *e, hipStream_t st, const int *A, int nsites, bool have_ycols) {
    dim3 g1(std::min(64, (e->wpc64 + WAVES - 1) / WAVES), nsites);
    if (!have_ycols) hipLaunchKernelGGL(tags_to_bits_kernel, g1, dim3(BLOCK), 0, st, A, e->strideA, e->M, e->ycols, e->wpc64);   // else: emitted by the maxWithin sweep
    const bool wide = e->wpc64 > 2048;                      // > 131072 haplotypes: 1024 threads per column
#ifdef PBWTAMD_MEASURE
    static const bool old_pack3 = tune_env("PBWTAMD_OLD_PACK3") != nullptr;   // the chunk-loop encoder (A/B runs)
    static const int p3_form = tune_env("PBWTAMD_PACK3_FORM") ? atoi(tune_env("PBWTAMD_PACK3_FORM")) : 3;   // 3 = region-parallel, 2 = one workgroup per column
#endif
#ifndef PBWTAMD_MEASURE
    launch_p3r_sizes(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes);
    (void)wide;
#else
    if (!old_pack3 && p3_form == 3) launch_p3r_sizes(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes);
    else if (!old_pack3) launch_pack3v2<0>(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, (uint8_t *)nullptr);
    else if (wide) hipLaunchKernelGGL((pack3_kernel<0, 1024>), dim3(nsites), dim3(1024), 0, st, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, (uint8_t *)nullptr);
    else hipLaunchKernelGGL((pack3_kernel<0>), dim3(nsites), dim3(BLOCK), 0, st, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, (uint8_t *)nullptr);
#endif
    // exclusive offsets inside the batch; batch total -> scal[2]
    hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, e->colBytes, (size_t)nsites, e->scal + 2, 0ULL);
    HIPCHK(hipGetLastError());
    const size_t worst = (size_t)nsites * (size_t)e->M;
    if (e->h_used) {                                       // newest asynchronous reading of the true count that has landed
        for (long long j = e->used_n - 1; j >= 0 && j >= e->used_n - 8; --j) {
            if (hipEventQuery(e->evUsed[j % 8]) != hipSuccess) continue;
            size_t later = 0;
            for (long long q = j + 1; q < e->used_n; ++q) later += e->usedWorstAfter[q % 8];
            e->yz_upper = std::min(e->yz_upper, (size_t)e->h_used[j % 8] + later);
            break;
        }
        (void)hipGetLastError();
    }
    if (e->yz_upper + worst > e->yzCap) {                  // refresh the bound with the true count, grow if needed
        unsigned long long used = 0;
        HIPCHK(hipMemcpyAsync(&used, e->scal + 1, sizeof used, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        e->yz_upper = (size_t)used;
        if (e->yz_upper + 8 * worst > e->yzCap) CHK(ensure_yz(e, st, std::max(e->yz_upper + 16 * worst, e->yzCap * 2)));
    }
    hipLaunchKernelGGL(pack3_offsets_kernel, dim3((nsites + 255) / 256), dim3(256), 0, st, e->colBytes, (size_t)nsites, (const unsigned long long *)(e->scal + 1),
                       e->scal + 2, e->scal + 1, (unsigned long long)e->yzCap, e->ctl + 2);
#ifndef PBWTAMD_MEASURE
    launch_p3r_emit(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes, e->yz);
#else
    if (!old_pack3 && p3_form == 3) launch_p3r_emit(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes, e->yz);
    else if (!old_pack3) launch_pack3v2<1>(st, nsites, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, e->yz);
    else if (wide) hipLaunchKernelGGL((pack3_kernel<1, 1024>), dim3(nsites), dim3(1024), 0, st, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, e->yz);
    else hipLaunchKernelGGL((pack3_kernel<1>), dim3(nsites), dim3(BLOCK), 0, st, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->colBytes, e->yz);
#endif
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, e->scal + 1, (const unsigned long long *)(e->scal + 2), (unsigned long long)e->yzCap, e->ctl + 2);
    HIPCHK(hipGetLastError());
    e->yz_upper += worst;
    if (!e->h_used) {
        HIPCHK(hipHostMalloc((void **)&e->h_used, 8 * sizeof(unsigned long long), hipHostMallocDefault));
        for (int i = 0; i < 8; ++i) HIPCHK(hipEventCreateWithFlags(&e->evUsed[i], hipEventDisableTiming));
    }
    {   // this batch's true running total, read back without waiting for it
        const int slot = (int)(e->used_n % 8);
        HIPCHK(hipMemcpyAsync(e->h_used + slot, e->scal + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(e->evUsed[slot], st));
        e->usedWorstAfter[slot] = worst;                   // bytes this batch may have added (for readings older than it)
        ++e->used_n;
    }
    return 0;
}

// first site of a pass (or after a mode switch / odd-length batch): tag slot 0 with the alleles of
// its site(s) and build the tile summaries from scratch, in the format of the step kernel to follow
static int ensure_prepared(pbwtamd_engine *e, const uint32_t *col, bool sorted, bool with_d, bool pair, int ncols_avail) {
    if (e->prepared && e->summ_pair == pair) return 0;
    if (pair) {
        Prep2Args p;
        p.a = ringA(e, e->ring); p.d = ringD(e, e->ring); p.col0 = col;
        p.col1 = (e->k_cur + 1 < e->n_total && ncols_avail > 1) ? col + e->wpc : e->zerocol;
        p.summ = e->summ; p.M = e->M; p.W = e->W; p.wpad = e->wpad; p.with_d = with_d;
        p.T = e->T;
        hipLaunchKernelGGL(prepare2_kernel, dim3(e->W), dim3(BLOCK), 0, e->stream, p);
    } else {
        PrepArgs p;
        p.a = ringA(e, e->ring); p.d = ringD(e, e->ring); p.col = col; p.summ = e->summ; p.k = e->k_cur; p.M = e->M; p.W = e->W;
        p.wpad = e->wpad; p.T = e->T; p.sorted = sorted; p.with_d = with_d; p.has_col = 1;
        hipLaunchKernelGGL(prepare_kernel, dim3(e->W), dim3(BLOCK), 0, e->stream, p);
    }
    HIPCHK(hipGetLastError());
    e->prepared = true; e->summ_cur = 0; e->summ_pair = pair;
    return 0;
}

// XCD-contiguous tile placement (xcd_tile): bit 0 fill, bit 1 rank, bit 2 hist.  Measured at M = 100 k: 1.600 -> 1.539 us/site.
static int xcd_flags() { static const int v = tune_env("PBWTAMD_XCD") ? atoi(tune_env("PBWTAMD_XCD")) : 7; return v; }

// batch consumers (checksums, maxWithin sweep, pack3) of the pending batch, on the second stream so
// they overlap the next batch's launch chain (which occupies only ~W of the 256 CUs)
// consumers over the sites kbase+j0 .. kbase+j0+ns-1 of batch p (slots j0 .. j0+ns-1 of its ring; j0, ns multiples of 8 on the skeleton path)
// what: bit 0 = the fill (and the query sweep's block summaries), on the consumer stream s2; bit 1 = everything that reads the filled states
// (checksums, maxWithin / longWithin sweeps, pack3), on stream sr
static int run_consumers(pbwtamd_engine *e, const Pending &p, int j0, int ns, int what = 3, hipStream_t sr = nullptr) {
    if (!sr) sr = e->s2;
    const int *A = ringA(e, p.ring) + (size_t)j0 * e->strideA, *D = ringD(e, p.ring) + (size_t)j0 * e->strideD;
    const int kb = p.kbase + j0;
    const bool with_d = p.opts & PBWTAMD_OPT_WITH_D;
#ifdef PBWTAMD_MEASURE                                     // measurement builds only (-DPBWTAMD_MEASURE): these switches give WRONG results
    static const bool nofill = getenv("PBWTAMD_NOFILL") && atoi(getenv("PBWTAMD_NOFILL"));
#else
    constexpr bool nofill = false;
#endif
    const unsigned consumers = PBWTAMD_OPT_CHECKSUM | PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_PACK3 | PBWTAMD_OPT_LONG_RECS | OPT_INTERNAL_KEEP_STATES;
    const bool packed = packed_fill(p);
    // the 16-bit hand-off (round 4; PBWTAMD_P16=0: the d | y << 31 slots): the sequential fill writes L | y << 15 into a ring of its own, the streaming
    // sweep reads that — half the bytes on both sides.  PBWTAMD_P16_CLIP=n (tests): lengths from n on escape to the 32-bit slot.
    // Measured (interleaved A/B, founder-mosaic panels, with the four-per-lane stores of the fill and the LDS-staged sweep): -9 % at 1 M haplotypes
    // (5.35 -> 4.87 us/site), -2.3..-2.6 % at 100 k, -3.4 % at 50 k; an iid panel (every level of every tile moves elements, every group of the sweep has
    // pending scans) +9 % at 100 k.  On at every width (PBWTAMD_P16=0: off).
    const bool p16_on = env_int("PBWTAMD_P16", 1) != 0 && env_int("PBWTAMD_FILL_SEQ", 1) != 0
#ifdef PBWTAMD_MEASURE
                               && !env_int("PBWTAMD_FILL_FUSE", 0) && !env_int("PBWTAMD_FILL_YC", 0) && !tune_env("PBWTAMD_OLD_SWEEP")
#endif
        ;
    const int p16_clip = std::min(std::max(env_int("PBWTAMD_P16_CLIP", P16_ESC), 1), (int)P16_ESC);
    const bool p16 = p16_on && packed && p.skel && e->skEPT <= 2;
    unsigned short *P16 = nullptr;
    if (p16) {
        if (!e->p16r) HIPCHK(dev_alloc((void **)&e->p16r, (size_t)2 * (e->B + 2) * e->strideD * sizeof(unsigned short)));
        P16 = e->p16r + ((size_t)p.ring * (e->B + 2) + j0) * e->strideD;
    }
    bool fused = false;                                     // this call's fill has decided most of the -stats sweep and emitted the bit columns
    bool yc = false;                                        // this call's fill has emitted the sorted allele columns (the sweep reads them, pack3 encodes them)
    if ((what & 1) && p.skel && !nofill && (p.opts & consumers)) {   // the 7 states between consecutive skeleton states: all blocks and tiles in one launch
        SkFillArgs f;
        f.A = ringA(e, p.ring) + (size_t)j0 * e->strideA; f.D = ringD(e, p.ring) + (size_t)j0 * e->strideD; f.strideA = e->strideA; f.strideD = e->strideD;
        f.keys = e->keysR[p.ring] + (size_t)(j0 / 8) * e->Mpad; f.strideK = e->Mpad; f.scan = e->saveR[p.ring] + (size_t)(j0 / 8) * e->strideS; f.strideS = e->strideS;
        f.M = e->M; f.W = e->Wt; f.kbase = kb;
#ifdef PBWTAMD_MEASURE
        static const int dbg_nowrite = getenv("PBWTAMD_DEBUG_FILL_NOWRITE") ? std::max(1, atoi(getenv("PBWTAMD_DEBUG_FILL_NOWRITE"))) : 0; f.dbg_nowrite = dbg_nowrite;
#endif
        f.pack_y = packed ? 1 : 0;
        f.xcd = xcd_flags() & 1;
        f.pair = e->prow ? 1 : 0; f.W2 = e->W2;
        dim3 grid(e->Wt, ns / 8);
        static const size_t dyn = tune_env("PBWTAMD_FILL_PAD_KB") ? (size_t)atoi(tune_env("PBWTAMD_FILL_PAD_KB")) * 1024 : 0;   // occupancy probe (results unchanged)
        const bool d_only = !packed && (p.opts & OPT_INTERNAL_D_ONLY);
#define FILL(EP) do { if (packed) hipLaunchKernelGGL((skel_fill_kernel<EP, 1>), grid, dim3(BLOCK), dyn, e->s2, f); \
                      else if (d_only) hipLaunchKernelGGL((skel_fill_kernel<EP, 2>), grid, dim3(BLOCK), dyn, e->s2, f); \
                      else hipLaunchKernelGGL((skel_fill_kernel<EP, 0>), grid, dim3(BLOCK), dyn, e->s2, f); } while (0)
        // the sequential tile-local form (pbwt_fillseq.h) whenever no consumer needs the haplotype ids; PBWTAMD_FILL_SEQ=0: the table form (A/B, bit-exact)
        const bool fill_seq = env_int("PBWTAMD_FILL_SEQ", 1) != 0;
        if (fill_seq && (packed || d_only) && e->skEPT <= 2) {
            SkFillPrepArgs pa; pa.scan = f.scan; pa.strideS = f.strideS; pa.nrow = e->prow ? e->W2 : e->Wt; pa.kbase = kb; pa.gb = e->fillGB[p.ring] + (size_t)(j0 / 8) * SKK;
            hipLaunchKernelGGL(skel_fillprep_kernel, dim3(ns / 8), dim3(BLOCK), 0, e->s2, pa);
            SkFillSeqArgs q; q.D = f.D; q.strideD = f.strideD; q.keys = f.keys; q.strideK = f.strideK; q.scan = f.scan; q.strideS = f.strideS; q.gb = pa.gb;
            q.M = e->M; q.W = e->Wt; q.kbase = kb; q.nblk = ns / 8; q.xcd = f.xcd; q.pair = f.pair; q.W2 = f.W2;
#ifdef PBWTAMD_MEASURE
            q.dbg_nowrite = f.dbg_nowrite;
#endif
            const dim3 gs(((size_t)e->Wt * (ns / 8) + WAVES - 1) / WAVES);
            // FUSED with the -stats sweep (PBWTAMD_FILL_FUSE=0: off): the fill decides the first step of matchMaximalWithin's scans for every position
            // whose neighbours stand in the same run, flags the rest for sweep_resid_kernel and emits the sorted bit columns pack3 encodes
#ifdef PBWTAMD_MEASURE
            // measurement builds only: built, bit-exact (every-position checksums, histogram, .pbwt bytes on mosaic and iid panels), and SLOWER — the fused
            // fill takes 1.85 ms per 512-site batch at 1 M haplotypes against 0.80 + 0.91 for fill + streaming sweep, the residual sweep 0.89 ms for the 1 %
            // of positions left to it: both consumers are bound by instruction issue, not by the bytes the fusion saves
            const bool fuse_env = env_int("PBWTAMD_FILL_FUSE", 0) != 0;
#else
            constexpr bool fuse_env = false;
#endif
            if (fuse_env) yc = false;
            fused = fuse_env && e->fuse_ok && packed && what == 3 && sr == e->s2 && (p.opts & PBWTAMD_OPT_WITHIN_HIST) && !(p.opts & PBWTAMD_OPT_WITHIN_RECS);
            q.flags = nullptr; q.strideF = 0; q.ycols = nullptr; q.wpc64 = e->wpc64; q.nflag = nullptr;
            // YC (PBWTAMD_FILL_YC=0: off): with the -stats sweep behind it the fill also emits every state's sorted allele column; the sweep reads those
            // first and pack3 encodes them
#ifdef PBWTAMD_MEASURE
            // measurement builds only (PBWTAMD_FILL_YC=1): bit-exact, and a wash — at 1 M haplotypes the fill goes from 0.65 to 0.79 ms per batch, the sweep from
            // 0.79 to 0.56 (it loads a quarter of the groups but its waves are then too short to hide their round trips), 5.47 -> 5.54 us/site end to end;
            // at 100 k 1.525 -> 1.486; on an iid panel (every chunk has ones, runs of a few positions: an atomic pair per run) 4.7 -> 6.0
            const bool yc_env = env_int("PBWTAMD_FILL_YC", 0) != 0;
#else
            constexpr bool yc_env = false;
#endif
            yc = yc_env && packed && what == 3 && sr == e->s2 && (p.opts & PBWTAMD_OPT_WITHIN_HIST) && !(p.opts & PBWTAMD_OPT_WITHIN_RECS);
            // (Two experiments on WHERE the fill's stores go, both bit-exact, both slower, both removed — DESIGN.md section 2: (1) the packed slots in a ring
            // of their own in UNCACHED device memory, so that the fill leaves no dirty lines in the L2s for the chain's kernel boundaries to write back:
            // the 4-byte-per-lane stores take 2.3x as long without the L2 to merge them (fill 0.65 -> 1.48 ms per batch at 1 M, 5.4 -> 6.9 us/site; the
            // sweep reads the uncached ring at the same 0.79 ms); (2) every n-th fill wave writing the L2 back itself when it is done (buffer_wbl2):
            // 5.29 us/site without, 5.42 / 5.69 / 6.50 / 9.13 with n = 256 / 64 / 16 / 4.)
            q.Dout = q.D; q.P16 = P16; q.stride16 = e->strideD; q.clip = p16_clip;
            if (yc && !fused) { q.ycols = e->ycols; HIPCHK(hipMemsetAsync(e->ycols, 0, (size_t)ns * e->wpc64 * sizeof(unsigned long long), e->s2)); }
            if (fused) {
                if (!e->wflags) {
                    e->strideF = (size_t)e->Mpad / 32;
                    HIPCHK(dev_alloc((void **)&e->wflags, (size_t)(e->B + 8) * e->strideF * sizeof(unsigned)));
                    HIPCHK(hipMemsetAsync(e->wflags, 0, (size_t)(e->B + 8) * e->strideF * sizeof(unsigned), e->s2));
                    HIPCHK(dev_alloc((void **)&e->nflag, sizeof(unsigned long long)));
                    HIPCHK(hipMemsetAsync(e->nflag, 0, sizeof(unsigned long long), e->s2));
                    HIPCHK(hipHostMalloc((void **)&e->h_nflag, sizeof(unsigned long long), hipHostMallocDefault)); *e->h_nflag = 0;
                    HIPCHK(hipEventCreateWithFlags(&e->evFlag, hipEventDisableTiming));
                }
                q.flags = e->wflags; q.strideF = e->strideF; q.nflag = e->nflag;
                if (p.opts & PBWTAMD_OPT_PACK3) { q.ycols = e->ycols; HIPCHK(hipMemsetAsync(e->ycols, 0, (size_t)ns * e->wpc64 * sizeof(unsigned long long), e->s2)); }
            }
#ifdef PBWTAMD_MEASURE
            if (fused) { if (e->skEPT == 1) hipLaunchKernelGGL((skel_fillseq_kernel<4, 1, 2>), gs, dim3(BLOCK), dyn, e->s2, q); else hipLaunchKernelGGL((skel_fillseq_kernel<8, 1, 2>), gs, dim3(BLOCK), dyn, e->s2, q); }
            else
#endif
#ifdef PBWTAMD_MEASURE
            if (yc) { if (e->skEPT == 1) hipLaunchKernelGGL((skel_fillseq_kernel<4, 1, 1>), gs, dim3(BLOCK), dyn, e->s2, q); else hipLaunchKernelGGL((skel_fillseq_kernel<8, 1, 1>), gs, dim3(BLOCK), dyn, e->s2, q); }
            else
#endif
            if (p16) { if (e->skEPT == 1) hipLaunchKernelGGL((skel_fillseq_kernel<4, 3>), gs, dim3(BLOCK), dyn, e->s2, q); else hipLaunchKernelGGL((skel_fillseq_kernel<8, 3>), gs, dim3(BLOCK), dyn, e->s2, q); }
            else
            if (e->skEPT == 1) { if (packed) hipLaunchKernelGGL((skel_fillseq_kernel<4, 1>), gs, dim3(BLOCK), dyn, e->s2, q); else hipLaunchKernelGGL((skel_fillseq_kernel<4, 2>), gs, dim3(BLOCK), dyn, e->s2, q); }
            // (77 VGPRs, 6 waves per SIMD.  Forced to 64 VGPRs / 8 waves — 12 registers spilled — the fill itself gains 4 % and the chain's rank launch
            // beside it goes from 13.0 to 19.9 us: 5.50 -> 5.74 us/site at 1 M.  Not kept.)
            else { if (packed) hipLaunchKernelGGL((skel_fillseq_kernel<8, 1>), gs, dim3(BLOCK), dyn, e->s2, q); else hipLaunchKernelGGL((skel_fillseq_kernel<8, 2>), gs, dim3(BLOCK), dyn, e->s2, q); }
        } else {
        static const bool fill_pair4 = tune_env("PBWTAMD_FILL_PAIR4") && atoi(tune_env("PBWTAMD_FILL_PAIR4"));   // measurement builds: with pair rows, one fill workgroup per PAIR (1024 positions, the pair's own scan row)
        if (fill_pair4 && e->prow && e->skEPT == 2) { f.W = e->W2; f.pair = 0; grid = dim3(e->W2, ns / 8); FILL(4); }
        else
        if (e->skEPT == 1) FILL(1); else if (e->skEPT == 2) FILL(2); else FILL(4);
        }
#undef FILL
        HIPCHK(hipGetLastError());
    }
    if ((what & 1) && e->qs_bsum[p.ring] && (p.opts & PBWTAMD_OPT_SORTED) && p.cols) {     // read side, for the query sweep: {max d, alleles present} per 256 positions of these states
        hipLaunchKernelGGL(qs_blocksum_kernel, dim3((e->qs_nblk + 4 * WAVES - 1) / (4 * WAVES), ns), dim3(BLOCK), 0, e->s2, D, e->strideD,
                           (const unsigned long long *)p.cols + (size_t)j0 * e->wpc64, e->wpc64, e->M, e->qs_nblk, e->qs_bsum[p.ring] + (size_t)j0 * e->qs_nblk);
        HIPCHK(hipGetLastError());
        if (j0 == e->qs_bsum_sites[p.ring]) e->qs_bsum_sites[p.ring] = j0 + ns;
    }
    if (!(what & 2)) return 0;
    if (p.opts & PBWTAMD_OPT_CHECKSUM) {
        unsigned long long *ca = e->csum + (kb - e->k0), *cd = ca + e->csum_sites, *cy = cd + e->csum_sites;
        dim3 grid(std::min(64, (e->M + BLOCK) / BLOCK), ns);
        hipLaunchKernelGGL(checksum_kernel, grid, dim3(BLOCK), 0, sr, A, D, e->strideA, e->strideD, e->M, with_d ? 1 : 0, ca, cd, cy, ns, packed ? (p16 ? 2 : 1) : 0, P16, e->strideD, kb);
        HIPCHK(hipGetLastError());
    }
#ifdef PBWTAMD_MEASURE
    if (fused) {                                            // what the fill left undecided: ~1 % of the positions of a founder-mosaic panel
        SweepResidArgs ra; ra.D = D; ra.strideD = e->strideD; ra.flags = e->wflags; ra.strideF = e->strideF; ra.M = e->M; ra.kbase = kb;
        ra.hist = e->hist; ra.histlen = e->histlen; ra.hist_rep = e->hist_rep; ra.err = e->ctl + 2;
        const int nwords = (e->M + 31) / 32;
        hipLaunchKernelGGL(sweep_resid_kernel, dim3((nwords + BLOCK - 1) / BLOCK, ns), dim3(BLOCK), 0, sr, ra);
        HIPCHK(hipGetLastError());
        // how much was left: read back without waiting; a panel whose scans rarely end at their first step (iid: half of all positions) is
        // cheaper through the streaming sweep, which reads every state once but tests 256 positions per wave and step
        if (e->flagPending && hipEventQuery(e->evFlag) == hipSuccess) {
            const unsigned long long tot = *e->h_nflag;
            const double frac = (double)(tot - e->nflag_prev) / std::max(1.0, e->flag_sites * (double)e->M);
            e->nflag_prev = tot; e->flag_sites = 0; e->flagPending = false;
            static const double fuse_max = getenv("PBWTAMD_FUSE_MAX_FLAGGED") ? atof(getenv("PBWTAMD_FUSE_MAX_FLAGGED")) : 0.10;
            if (frac > fuse_max) e->fuse_ok = false;
        }
        (void)hipGetLastError();
        e->flag_sites += ns;
        if (!e->flagPending) {
            HIPCHK(hipMemcpyAsync(e->h_nflag, e->nflag, sizeof(unsigned long long), hipMemcpyDeviceToHost, sr));
            HIPCHK(hipEventRecord(e->evFlag, sr));
            e->flagPending = true;
        }
    } else
#endif
    if (p.opts & (PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS)) CHK(run_within(e, sr, A, D, kb, ns, -1, p.opts, packed, yc, P16, p16_clip));
    if (p.opts & PBWTAMD_OPT_LONG_RECS) {
        CHK(run_long(e, sr, A, D, nullptr, kb, ns, -1));
        // keep the batch's last state (before site kbase+nb-1): it is the stale allele column if the panel ends here
        if (!e->ystale) HIPCHK(dev_alloc((void **)&e->ystale, sizeof(int) * e->strideA));
        HIPCHK(hipMemcpyAsync(e->ystale, A + (size_t)(ns - 1) * e->strideA, sizeof(int) * e->strideA, hipMemcpyDeviceToDevice, sr));
    }
    static const bool no_fuse = tune_env("PBWTAMD_NO_YCOLS_FUSION") != nullptr;
    if (p.opts & PBWTAMD_OPT_PACK3) CHK(run_pack3(e, sr, A, ns, fused || yc || (!no_fuse && (p.opts & PBWTAMD_OPT_WITHIN_HIST) != 0)));
    return 0;
}

static int shard_flush_pending(pbwtamd_engine *e);

static int flush_pending(pbwtamd_engine *e) {
    if (!e->pend.valid) return 0;
    if (e->sh) return shard_flush_pending(e);
    const Pending p = e->pend;
    e->pend.valid = false;
    // evRounds: everything the consumers READ is there while the batch's last rank launch may still be waiting for the other
    // ring — not for the packed fill, which rewrites the skeleton slots' d in place (d | y << 31), the last round's input among them
    HIPCHK(hipStreamWaitEvent(e->s2, (p.early && !packed_fill(p)) ? e->evRounds[p.ring] : e->evChain[p.ring], 0));
    // (measured, not kept: fill on s2 and sweep + pack3 on a third stream in sub-batches of 64-256 sites, so that the two run beside each other —
    // 5.81 -> 5.97 us/site at 1 M, 1.68 -> 1.83 at 100 k: end to end is the CHAIN's time beside the consumers, not the consumers' own)
    if (p.nb > p.flushed) CHK(run_consumers(e, p, p.flushed, p.nb - p.flushed));
    HIPCHK(hipEventRecord(e->evCons[p.ring], e->s2));
    e->consRecorded[p.ring] = true;
    return 0;
}

// one batch of the skeleton chain: nb (multiple of 8) sites from slot 0 of ring r, writing slots 8, 16, ..., nb.
// cols: the batch's bit columns (navail of them, original order)
static inline bool skel_two_launch(const pbwtamd_engine *e) { return e->skn && e->Wt <= e->skn_maxw; }

// launch helpers for the skeleton kernels: EPT = positions per thread (tile = 256*EPT)
// hist + the per-key scan over the tiles, into g.scan / g.total (and g.tbl0 with pair rows) on stream st; returns true when the
// wide (two-level) scan ran.  agg / cnt / epoch: the two-level scan's aggregates and arrival counter (one set per stream).
template <int EPT>
static bool launch_skel_hist_scan(pbwtamd_engine *e, hipStream_t st, const SkArgs &g, unsigned long long *agg, unsigned *cnt, unsigned *epoch) {
    const int W = g.W;
    if (e->prow) {                                         // wide panels: hist and scan on PAIRS of tiles (half the rows), rank on tiles
        SkArgs h = g; h.W = e->W2; h.Wtot = e->W2;
        if (EPT == 1) hipLaunchKernelGGL((skel_hist_kernel<2, true>), dim3(e->W2), dim3(BLOCK), 0, st, h);
        else hipLaunchKernelGGL((skel_hist_kernel<4, true>), dim3(e->W2), dim3(BLOCK), 0, st, h);
        static const int one_max = tune_env("PBWTAMD_PROW_ONE_MAX") ? std::min(1024, atoi(tune_env("PBWTAMD_PROW_ONE_MAX"))) : 512;
        if (e->W2 <= one_max) {                            // few enough rows for the one-level scan
            Sk2Args k2; k2.tbl = g.tbl; k2.scan = g.scan; k2.total = g.total; k2.W = e->W2;
            static const int lean = tune_env("PBWTAMD_K2_LEAN") ? atoi(tune_env("PBWTAMD_K2_LEAN")) : 0;   // measurement builds: two keys per workgroup (17 KB of LDS)
            if (lean && e->W2 > 512) hipLaunchKernelGGL((skel_k2_kernel<2, 16>), dim3(SKK / 2), dim3(128), 0, st, k2);
            else if (lean && e->W2 > 256) hipLaunchKernelGGL((skel_k2_kernel<2, 8>), dim3(SKK / 2), dim3(128), 0, st, k2);
            else
            if (e->W2 <= 256) hipLaunchKernelGGL((skel_k2_kernel<4, 4>), dim3(SKK / 4), dim3(BLOCK), 0, st, k2);
            else hipLaunchKernelGGL((skel_k2_kernel<4, 16>), dim3(SKK / 4), dim3(BLOCK), 0, st, k2);
            return false;
        }
        Sk2WArgs kw; kw.tbl = g.tbl; kw.scan = g.scan; kw.total = g.total; kw.W = e->W2; kw.agg = agg; kw.counter = cnt; kw.err = e->ctl + 2;
        static const int tpw_env = tune_env("PBWTAMD_K2_TPW") ? atoi(tune_env("PBWTAMD_K2_TPW")) : 32;     // rows per workgroup (measurement builds: 16 / 32; <= 64 workgroups)
        const int tpw = e->W2 > 2048 ? 64 : tpw_env;
        const int nwg = (e->W2 + tpw - 1) / tpw;
        *epoch += (unsigned)nwg; kw.target = *epoch;
        static const int pch = tune_env("PBWTAMD_K2_PCH") ? atoi(tune_env("PBWTAMD_K2_PCH")) : 32;        // measurement builds: aggregates in flight per lane (16: 48 VGPRs instead of 74)
        if (pch == 16 && tpw == 32) hipLaunchKernelGGL((skel_k2_wide_kernel<32, 16, 16>), dim3(nwg), dim3(SKK), 0, st, kw);
        else if (pch == 16 && tpw == 64) hipLaunchKernelGGL((skel_k2_wide_kernel<64, 16, 16>), dim3(nwg), dim3(SKK), 0, st, kw);
        else if (pch == 8 && tpw == 32) hipLaunchKernelGGL((skel_k2_wide_kernel<32, 8, 8>), dim3(nwg), dim3(SKK), 0, st, kw);
        else
        if (tpw == 64) hipLaunchKernelGGL((skel_k2_wide_kernel<64>), dim3(nwg), dim3(SKK), 0, st, kw);
        else if (tpw == 16 && nwg <= 64) hipLaunchKernelGGL((skel_k2_wide_kernel<16, 16, 32>), dim3(nwg), dim3(SKK), 0, st, kw);
        else hipLaunchKernelGGL((skel_k2_wide_kernel<32>), dim3(nwg), dim3(SKK), 0, st, kw);
        return true;
    }
    hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, st, g);
    static const bool k2_wide = !(tune_env("PBWTAMD_K2_WIDE") && !atoi(tune_env("PBWTAMD_K2_WIDE")));
    if ((W > 512 && k2_wide) || W > 2048) {                // two-level scan in one launch: <= 64 co-resident workgroups of 32 (64) tiles
        Sk2WArgs kw; kw.tbl = g.tbl; kw.scan = g.scan; kw.total = g.total; kw.W = W; kw.agg = agg; kw.counter = cnt; kw.err = e->ctl + 2;
        const int tpw = W > 2048 ? 64 : 32, nwg = (W + tpw - 1) / tpw;
        *epoch += (unsigned)nwg; kw.target = *epoch;
        static const int pch = tune_env("PBWTAMD_K2_PCH") ? atoi(tune_env("PBWTAMD_K2_PCH")) : 32;        // measurement builds: aggregates in flight per lane (16: 48 VGPRs instead of 74)
        if (pch == 16 && tpw == 32) hipLaunchKernelGGL((skel_k2_wide_kernel<32, 16, 16>), dim3(nwg), dim3(SKK), 0, st, kw);
        else if (pch == 16 && tpw == 64) hipLaunchKernelGGL((skel_k2_wide_kernel<64, 16, 16>), dim3(nwg), dim3(SKK), 0, st, kw);
        else if (pch == 8 && tpw == 32) hipLaunchKernelGGL((skel_k2_wide_kernel<32, 8, 8>), dim3(nwg), dim3(SKK), 0, st, kw);
        else
        if (tpw == 64) hipLaunchKernelGGL((skel_k2_wide_kernel<64>), dim3(nwg), dim3(SKK), 0, st, kw);
        else hipLaunchKernelGGL((skel_k2_wide_kernel<32>), dim3(nwg), dim3(SKK), 0, st, kw);
        return true;
    }
    Sk2Args k2; k2.tbl = g.tbl; k2.scan = g.scan; k2.total = g.total; k2.W = W;
    if (W <= 256) hipLaunchKernelGGL((skel_k2_kernel<4, 4>), dim3(SKK / 4), dim3(BLOCK), 0, st, k2);   // one key per wave (16 / 8 keys per workgroup measured slower)
    else if (W <= 1024) hipLaunchKernelGGL((skel_k2_kernel<4, 16>), dim3(SKK / 4), dim3(BLOCK), 0, st, k2);
    else hipLaunchKernelGGL((skel_k2_kernel<2, 32>), dim3(SKK / 2), dim3(128), 0, st, k2);
    return false;
}

// part: 0 = the whole round; 1 = hist + tile scan only; 2 = the rank launch only (after a part-1 call with the same arguments).
// Rounds of two launches (the rank scans the tile table itself) cannot be split: part 1 does nothing, part 2 the whole round.
template <int EPT>
static void launch_skel_round(pbwtamd_engine *e, SkArgs &g, bool two_launch, int part = 0) {
    const int W = g.W;
    if (two_launch && !e->prow) {
        if (part == 1) return;
        hipLaunchKernelGGL((skel_hist_kernel<EPT>), dim3(W), dim3(BLOCK), 0, e->stream, g);
        if (W <= 16) hipLaunchKernelGGL((skel_rank_kernel<EPT, 16>), dim3(W), dim3(BLOCK), 0, e->stream, g);
        else if (W <= 32) hipLaunchKernelGGL((skel_rank_kernel<EPT, 32>), dim3(W), dim3(BLOCK), 0, e->stream, g);
        else if (W <= 64) hipLaunchKernelGGL((skel_rank_kernel<EPT, 64>), dim3(W), dim3(BLOCK), 0, e->stream, g);
        else hipLaunchKernelGGL((skel_rank_kernel<EPT, SKN_MAXW>), dim3(W), dim3(BLOCK), 0, e->stream, g);
        return;
    }
    static const bool k2_wide_on = !(tune_env("PBWTAMD_K2_WIDE") && !atoi(tune_env("PBWTAMD_K2_WIDE")));
    static const int one_max_r = tune_env("PBWTAMD_PROW_ONE_MAX") ? std::min(1024, atoi(tune_env("PBWTAMD_PROW_ONE_MAX"))) : 512;
    bool wide = (e->prow && e->W2 > one_max_r) || (!e->prow && W > 512 && k2_wide_on) || W > 2048;
    if (part != 2) wide = launch_skel_hist_scan<EPT>(e, e->stream, g, e->k2agg, e->k2cnt, &e->k2epoch);
    if (part == 1) return;
    static const bool rank_r4 = !(tune_env("PBWTAMD_RANK_R4") && !atoi(tune_env("PBWTAMD_RANK_R4")));
    // the 22 KB rank workgroup (radix-4 range maxima) from two workgroups per CU on, whatever the scan: end to end 2.68 -> 2.56 us/site at
    // 300 k haplotypes (586 tiles), 3.13 -> 2.94 at 400 k; nothing up to 250 k (489 tiles)
    // (round 4) the radix-4 form at EVERY width of the three-launch round: 20 KB of LDS instead of 28.7.  Alone it is as fast as the radix-2 form below 512 tiles
    // (1.495 against 1.499 us/site at 100 k), and beside the sweep of the 16-bit slots (20.5 KB per workgroup, seven per CU) it fits the hole a retiring sweep
    // workgroup leaves, which the 28.7 KB workgroup does not (section 2's rule).
    static const int r4_from = tune_env("PBWTAMD_RANK_R4_FROM") ? atoi(tune_env("PBWTAMD_RANK_R4_FROM")) : 0;
    if ((wide && (e->prow || rank_r4)) || W >= r4_from) hipLaunchKernelGGL((skel_rank_kernel<EPT, 0, true>), dim3(W), dim3(BLOCK), 0, e->stream, g);    // more tiles than one round of the chip: occupancy counts
    else hipLaunchKernelGGL((skel_rank_kernel<EPT, 0>), dim3(W), dim3(BLOCK), 0, e->stream, g);
}

// the skeleton chain of one batch: slot 8s -> slot 8s+8 with an 8-bit radix step (keys = the alleles
// at the 8 sites, gathered through the transposed panel and carried along with the state).
// skel_prepare: transposed panel of the batch and the keys of slot 0 (unless the previous batch's last round delivered them).
static void skel_transpose(pbwtamd_engine *e, hipStream_t st, uint32_t *xT, const uint32_t *cols, int nb, int nvalid) {
    const int nblk = (std::min(nb + 8, nvalid) + 31) / 32;
    dim3 gt((e->wpc + BLOCK - 1) / BLOCK, nblk);
    hipLaunchKernelGGL(transpose32_kernel, gt, dim3(BLOCK), 0, st, cols, e->wpc, nvalid, xT, e->strideX, e->Mpad);
}

static int skel_prepare(pbwtamd_engine *e, int r, const uint32_t *cols, int nb, int navail, bool sorted) {
    const int nvalid = std::min(navail, e->n_total - e->k_cur);
    if (sorted) {                                          // read side: keys of every round from the sorted columns (LF-mapping), slot 0 tagged by position
        const unsigned long long *yc = (const unsigned long long *)cols;
        if (e->evPreKeys) {                                 // decoded columns and keysR[r] were prepared ahead of the chain (pbwtamd_match_sweep_sparse)
            HIPCHK(hipStreamWaitEvent(e->stream, e->evPreKeys, 0));
            e->evPreKeys = nullptr;
        } else {
            hipLaunchKernelGGL(qs_rankdir_kernel, dim3(nb), dim3(BLOCK), 0, e->stream, yc, e->wpc64, e->M, e->rankdirS);
            hipLaunchKernelGGL(skel_keys_sorted_kernel, dim3((e->M + BLOCK - 1) / BLOCK, nb / 8), dim3(BLOCK), 0, e->stream, yc, e->wpc64,
                               (const int *)e->rankdirS, e->M, e->keysR[r], (size_t)e->Mpad);
        }
        if (!e->keys_ready[r]) hipLaunchKernelGGL(skel_tag_sorted_kernel, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, ringA(e, r), yc, e->M);
    } else {
        skel_transpose(e, e->stream, e->xTr[r], cols, nb, nvalid);
        if (!e->keys_ready[r])
            hipLaunchKernelGGL(skel_keys_kernel, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, ringA(e, r), (const unsigned char *)e->xTr[r], e->M, e->keysR[r]);
    }
    e->keys_ready[r] = false;
    HIPCHK(hipGetLastError());
    return 0;
}

// rounds [s_from, s_to) of the batch; `direct`: the batch's last round scatters straight into slot 0
// (and the slot-0 keys) of the other ring, where the next batch starts
// the arguments of round s8 of a skeleton batch on ring r
static SkArgs skel_round_args(pbwtamd_engine *e, int r, const uint32_t *cols, bool sorted, int nb, int navail, int s8, bool direct) {
    int *A = ringA(e, r), *D = ringD(e, r);
    const int nvalid = std::min(navail, e->n_total - e->k_cur);
    unsigned char *kb = e->keysR[r];
    const uint32_t *xT = e->xTr[r];
    const int W = e->Wt;
    SkArgs g;
    g.tbl = (int2 *)e->skT;
    g.M = e->M; g.W = W; g.xcd = xcd_flags(); g.w0 = 0; g.Wtot = W;
    const int site = 8 * s8;                               // relative to the batch
    const bool last = direct && s8 == nb / 8 - 1;
    g.a = A + (size_t)site * e->strideA; g.d = D + (size_t)site * e->strideD; g.keys = kb + (size_t)s8 * e->Mpad;
    g.a_out = last ? ringA(e, r ^ 1) : A + (size_t)(site + 8) * e->strideA;
    g.d_out = last ? ringD(e, r ^ 1) : D + (size_t)(site + 8) * e->strideD;
    g.keys_out = last ? e->keysR[r ^ 1] : kb + (size_t)(s8 + 1) * e->Mpad;
    int2 *sv = e->saveR[r] + (size_t)s8 * e->strideS;      // this round's per-key scan over the tiles, kept for the fill
    const size_t nrow = e->prow ? (size_t)e->W2 : (size_t)W;
    g.scan = sv; g.total = reinterpret_cast<int *>(sv + nrow * SKK); g.tbl0 = sv + nrow * SKK + SKK / 2; g.pair = e->prow ? 1 : 0;
    g.has_next = (e->k_cur + site + 8 < e->n_total) && (site + 8 < nvalid);
    g.kbnext = reinterpret_cast<const unsigned char *>(xT) + (size_t)((site + 8) / 8) * e->strideX;   // byte plane of sites site+8 .. site+15
    g.ycnext = sorted ? (const unsigned long long *)cols + (size_t)(site + 8) * e->wpc64 : nullptr;
    g.k = e->k_cur + site;
    return g;
}

// all rounds of the batch in one launch (e->persist; two-launch regime): the last round writes slot nb of the SAME ring — the caller
// carries it into the other ring once that ring's readers are done
static int skel_rounds_persistent(pbwtamd_engine *e, int r, const uint32_t *cols, bool sorted, int nb, int navail) {
    const int nr = nb / 8, maxr = e->B / 8 + 1;
    if (!e->pargs) {
        HIPCHK(dev_alloc((void **)&e->pargs, 2 * (size_t)maxr * sizeof(SkArgs)));
        HIPCHK(hipHostMalloc((void **)&e->pargs_host, 2 * (size_t)maxr * sizeof(SkArgs), hipHostMallocDefault));
        HIPCHK(dev_alloc((void **)&e->pbar, 64)); HIPCHK(hipMemsetAsync(e->pbar, 0, 64, e->stream)); e->pbar_epoch = 0;
        for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&e->evPargs[i], hipEventDisableTiming));
    }
    const int h = e->pargs_half; e->pargs_half ^= 1;
    HIPCHK(hipEventSynchronize(e->evPargs[h]));            // the copy out of this half of the pinned staging (two batches ago) is done
    SkArgs *host = e->pargs_host + (size_t)h * maxr, *dev = e->pargs + (size_t)h * maxr;
    for (int s8 = 0; s8 < nr; ++s8) host[s8] = skel_round_args(e, r, cols, sorted, nb, navail, s8, false);
    HIPCHK(hipMemcpyAsync(dev, host, (size_t)nr * sizeof(SkArgs), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->evPargs[h], e->stream));
    const int W = e->Wt;
    const unsigned base = e->pbar_epoch;
    e->pbar_epoch += 2u * (unsigned)nr * (unsigned)W;
#define PERSIST(EP, TRR) hipLaunchKernelGGL((skel_persist_kernel<EP, TRR>), dim3(W), dim3(BLOCK), 0, e->stream, (const SkArgs *)dev, nr, e->pbar, base, e->ctl + 2)
#define PERSIST_TR(EP) do { if (W <= 16) PERSIST(EP, 16); else if (W <= 32) PERSIST(EP, 32); else if (W <= 64) PERSIST(EP, 64); else PERSIST(EP, SKN_MAXW); } while (0)
    if (e->skEPT == 1) PERSIST_TR(1); else if (e->skEPT == 2) PERSIST_TR(2); else PERSIST_TR(4);
#undef PERSIST_TR
#undef PERSIST
    HIPCHK(hipGetLastError());
    return 0;
}

static int skel_rounds(pbwtamd_engine *e, int r, const uint32_t *cols, bool sorted, int nb, int navail, int s_from, int s_to, bool direct, int part = 0) {
    const bool two = skel_two_launch(e);
    for (int s8 = s_from; s8 < s_to; ++s8) {
        const bool last = direct && s8 == nb / 8 - 1;
        SkArgs g = skel_round_args(e, r, cols, sorted, nb, navail, s8, direct);
        if (e->skEPT == 1) launch_skel_round<1>(e, g, two, part); else if (e->skEPT == 2) launch_skel_round<2>(e, g, two, part); else launch_skel_round<4>(e, g, two, part);
        if (part == 1) continue;
        if (last) e->keys_ready[r ^ 1] = g.has_next != 0;
        if (e->thr_rounds > 0 && (s8 + 1) % e->thr_rounds == 0) {   // a deep command queue slows the dependent chain down (measured): stay just ahead
            HIPCHK(hipEventRecord(e->tev[e->tev_n % 16], e->stream));
            if (e->tev_n >= e->thr_depth) HIPCHK(hipEventSynchronize(e->tev[(e->tev_n - e->thr_depth) % 16]));
            ++e->tev_n;
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

#include "pbwt_shard.inc"

extern "C" int pbwtamd_pass_advance(pbwtamd_engine *e, const void *d_bitcols, int wpc, int ncols, int ncols_avail, unsigned opts) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->pass_open) return fail("pbwtamd_pass_advance without pass_begin");
    if (wpc != e->wpc) return fail("pbwtamd_pass_advance: wpc %d != engine wpc %d", wpc, e->wpc);
    if (e->k_cur + ncols > e->n_total) return fail("pbwtamd_pass_advance: beyond n_total");
    if (ncols_avail < ncols + 1 && e->k_cur + ncols < e->n_total)
        return fail("pbwtamd_pass_advance: need the column after the batch (ncols_avail >= ncols+1) except at the last site");
    const bool with_d = opts & PBWTAMD_OPT_WITH_D, sorted = opts & PBWTAMD_OPT_SORTED;
    if ((opts & (PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_LONG_RECS)) && !with_d)
        return fail("pbwtamd: the maxWithin sweep needs OPT_WITH_D");
    if ((opts & (PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS)) && e->M < 2)
        return fail("pbwtamd: the maxWithin sweep needs at least 2 haplotypes (the reference reads y[-1] for M = 1)");
    const uint32_t *cols = (const uint32_t *)d_bitcols;
    int done = 0;
    while (done < ncols) {
        const int nb = std::min(e->B, ncols - done);
        const uint32_t *bc = cols + (size_t)done * wpc;
        const int r = e->ring;
        e->qs_bsum_sites[r] = 0;
        int *A = ringA(e, r), *D = ringD(e, r);
        // two sites per launch when the columns are in original order (the keys of the next pair are
        // gathered by haplotype) and the pair's successor columns are at hand
        const int left = ncols_avail - done;               // columns available from bc on
        const int remaining = e->n_total - e->k_cur;
        const int L = (nb + 1) / 2;
        const bool skel_read = env_int("PBWTAMD_SKEL_READ", 1) != 0;
        // the skeleton always carries d (A-only passes run it too: the divergences cost nothing on its critical path)
        const bool skel = e->skel && (nb % 8 == 0) && (sorted ? skel_read && left >= std::min(nb + 1, remaining) : left >= std::min(nb + 8, remaining));
        const bool pair = !skel && e->pair && !sorted && (e->T == 256 || e->T == 1024 || e->T == 2048 || e->T == 4096) && left >= std::min(2 * L + 2, remaining);
        if (e->sh) {                                       // one rank of a position-sharded panel (pbwt_shard.inc)
            if (sorted) return fail("pbwtamd: a position-sharded engine runs the build side only (original-order columns)");
            if (!e->sh->connected) return fail("pbwtamd: pbwtamd_shard_connect has not been called");
            if (skel) { CHK(shard_batch(e, bc, nb, left, opts)); done += nb; continue; }
            CHK(shard_make_full(e));                       // a batch the skeleton cannot take: replicated on every rank
        }
        if (!skel) {
            CHK(ensure_prepared(e, bc, sorted, with_d, pair, left));
            hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(256), 0, e->stream, e->ctlblk, e->k_cur, e->n_total, bc, (const uint32_t *)e->zerocol,
                               e->summ, pair ? 3 * e->wpad : e->wpad, e->summ_cur);
            e->keys_ready[r] = false;
        }
        const int nlaunch = skel ? ((e->persist && skel_two_launch(e) && !e->prow) ? 1 : (skel_two_launch(e) ? 2 : 3) * (nb / 8)) : (pair ? L : nb);
        if (!skel) e->summ_cur = nlaunch % 3;
        HIPCHK(hipGetLastError());
        // ---- the chain: slot j -> slot j+1 (-> slot j+2) of ring r ----
        if (e->ev_used == e->ev.size()) {
            hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); e->ev.push_back({a, b});
        }
        HIPCHK(hipEventRecord(e->ev[e->ev_used].first, e->stream));
        bool launched = false, early = false;
        int flushed_sites = 0;
        if (skel) {
            // ring r's consumers (incl. the fill that read xTr[r], keysR[r], saveR[r]) were waited for before slot 0 of ring r was written
            e->xT = e->xTr[r];
            CHK(skel_prepare(e, r, bc, nb, left, sorted));
            // the other stream's work is enqueued once all but the last round of this batch are (measured: better than right away)
            static const int flush_at = tune_env("PBWTAMD_FLUSH_AT") ? atoi(tune_env("PBWTAMD_FLUSH_AT")) : -1;   // rounds enqueued before the consumers (-1: all but the last)
            const int nr = nb / 8;
            if (e->persist && skel_two_launch(e) && !e->prow) {    // a small panel beside a wide one: the whole batch's chain in one launch
                CHK(skel_rounds_persistent(e, r, bc, sorted, nb, left));
                CHK(flush_pending(e));
                if (e->consRecorded[r ^ 1]) HIPCHK(hipStreamWaitEvent(e->stream, e->evCons[r ^ 1], 0));
                HIPCHK(hipMemcpyAsync(ringA(e, r ^ 1), A + (size_t)nb * e->strideA, sizeof(int) * e->strideA, hipMemcpyDeviceToDevice, e->stream));
                HIPCHK(hipMemcpyAsync(ringD(e, r ^ 1), D + (size_t)nb * e->strideD, sizeof(int) * e->strideD, hipMemcpyDeviceToDevice, e->stream));
                e->keys_ready[r ^ 1] = false;
                launched = true; e->prepared = false;
            } else {
            int s_done = 0;
            const unsigned cons_mask = PBWTAMD_OPT_CHECKSUM | PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_PACK3 | OPT_INTERNAL_KEEP_STATES;
            if (e->sub_rounds > 0 && nr > e->sub_rounds && (opts & cons_mask) && !(opts & (PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_LONG_RECS))) {
                // sub-batches: the consumers of rounds [s, s + sub_rounds) go to the consumer stream as soon as those rounds are enqueued — the
                // fill of the first rounds runs beside the chain of the later ones instead of after the whole batch
                CHK(flush_pending(e));                     // the previous batch's consumers come first on that stream
                Pending pt; pt.valid = true; pt.ring = r; pt.kbase = e->k_cur; pt.nb = nb; pt.opts = opts; pt.skel = true; pt.cols = bc;
                while (s_done + e->sub_rounds <= nr - 1) {
                    CHK(skel_rounds(e, r, bc, sorted, nb, left, s_done, s_done + e->sub_rounds, true));
                    hipEvent_t &evs = e->evSub[e->evSub_n++ % 8];
                    if (!evs) HIPCHK(hipEventCreateWithFlags(&evs, hipEventDisableTiming));
                    HIPCHK(hipEventRecord(evs, e->stream));
                    HIPCHK(hipStreamWaitEvent(e->s2, evs, 0));
                    CHK(run_consumers(e, pt, 8 * s_done, 8 * e->sub_rounds));
                    s_done += e->sub_rounds;
                }
                flushed_sites = 8 * s_done;
            }
            const int head = std::max(s_done, (flush_at >= 0) ? std::min(flush_at, nr - 1) : nr - 1);
            CHK(skel_rounds(e, r, bc, sorted, nb, left, s_done, head, true));
            // consumers of the PREVIOUS batch (other ring) are enqueued now, beside this batch's chain
            CHK(flush_pending(e));
            // the last round scatters straight into slot 0 of the other ring, once its readers are done
            static const bool no_direct = tune_env("PBWTAMD_NO_DIRECT") != nullptr;
            CHK(skel_rounds(e, r, bc, sorted, nb, left, head, nr - 1, !no_direct));
            // the last round's hist + tile scan read this ring only: with them everything this batch's consumers need is done
            // (evRounds) — only its rank launch, which scatters into the OTHER ring, has to wait for that ring's consumers
            CHK(skel_rounds(e, r, bc, sorted, nb, left, nr - 1, nr, !no_direct, 1));
            early = !skel_two_launch(e) || e->prow;
            if (early) { HIPCHK(hipEventRecord(e->evRounds[r], e->stream)); e->roundsRecorded[r] = true; }
            if (e->consRecorded[r ^ 1]) HIPCHK(hipStreamWaitEvent(e->stream, e->evCons[r ^ 1], 0));
            CHK(skel_rounds(e, r, bc, sorted, nb, left, nr - 1, nr, !no_direct, 2));
            if (no_direct) {
                HIPCHK(hipMemcpyAsync(ringA(e, r ^ 1), A + (size_t)nb * e->strideA, sizeof(int) * e->strideA, hipMemcpyDeviceToDevice, e->stream));
                HIPCHK(hipMemcpyAsync(ringD(e, r ^ 1), D + (size_t)nb * e->strideD, sizeof(int) * e->strideD, hipMemcpyDeviceToDevice, e->stream));
            }
            launched = true; e->prepared = false;
            }
        }
        else if (e->use_graph && nb == e->B) {
            hipGraphExec_t exec;
            if (get_graph(e, with_d, sorted, r, pair, &exec) == 0 && hipGraphLaunch(exec, e->stream) == hipSuccess) launched = true;
            else {                                         // capture / instantiate / launch refused: fall back to eager launches for good
                (void)hipGetLastError();
                e->use_graph = false;
                fprintf(stderr, "pbwt_amd: hipGraph path unavailable (%s); using eager launches\n", g_err.c_str());
            }
        }
        if (!launched) {
            if (pair) { for (int jl = 0; jl < L; ++jl) launch_step2(e, r, jl, with_d); }
            else { for (int j = 0; j < nb; ++j) launch_step_dyn(e, r, j, with_d, sorted); }
            HIPCHK(hipGetLastError());
        }
        if (pair && (nb & 1)) e->prepared = false;         // slot nb is a level-1 output: re-derive tags and summaries
        HIPCHK(hipEventRecord(e->ev[e->ev_used].second, e->stream));
        HIPCHK(hipEventRecord(e->evChain[r], e->stream)); e->chainRecorded[r] = true;
        ++e->ev_used; e->launches += nlaunch; e->sites_done += nb;
        if (!skel) {
            // ---- consumers of the PREVIOUS batch (other ring) run now, beside this batch's chain ----
            CHK(flush_pending(e));
            // ---- carry the cursor into slot 0 of the other ring once its readers are done ----
            if (e->consRecorded[r ^ 1]) HIPCHK(hipStreamWaitEvent(e->stream, e->evCons[r ^ 1], 0));
            if (e->sh) shard_wait_ring(e);                 // peers may still be pulling the other ring's slots
            HIPCHK(hipMemcpyAsync(ringA(e, r ^ 1), A + (size_t)nb * e->strideA, sizeof(int) * e->strideA, hipMemcpyDeviceToDevice, e->stream));
            if (with_d) HIPCHK(hipMemcpyAsync(ringD(e, r ^ 1), D + (size_t)nb * e->strideD, sizeof(int) * e->strideD, hipMemcpyDeviceToDevice, e->stream));
        }
        if (skel || (opts & (PBWTAMD_OPT_CHECKSUM | PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_PACK3 | PBWTAMD_OPT_LONG_RECS))) {
            e->pend.valid = true; e->pend.ring = r; e->pend.kbase = e->k_cur; e->pend.nb = nb; e->pend.opts = opts; e->pend.skel = skel; e->pend.sharded = false; e->pend.early = early; e->pend.flushed = flushed_sites; e->pend.cols = bc;
        }
        e->ring_skel[r] = skel;
        e->ring = r ^ 1;
        e->k_cur += nb;
        done += nb;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- many panels per launch
// P engines of the same width and batch, created on the SAME stream, advanced in lockstep: every chain launch covers all P panels
// (grid.y = panel, skel_*_many_kernel).  A chain launch below ~250 k haplotypes costs its 3-4 us whatever it does, so
// P panels — the chromosomes of one cohort — share that cost.  Consumers (fill, sweeps, pack3) stay per engine, on each engine's own
// consumer stream.  Batches the skeleton cannot take, wide (pair-row / two-level-scan) panels and special modes fall back to one
// pbwtamd_pass_advance per engine.
template <int EPT>
static void launch_round_many(pbwtamd_engine *e0, const SkArgs *dargs, int P, int W, bool two) {
    hipStream_t st = e0->stream;
    hipLaunchKernelGGL((skel_hist_many_kernel<EPT>), dim3(W, P), dim3(BLOCK), 0, st, dargs);
    if (two) {
        if (W <= 16) hipLaunchKernelGGL((skel_rank_many_kernel<EPT, 16>), dim3(W, P), dim3(BLOCK), 0, st, dargs);
        else if (W <= 32) hipLaunchKernelGGL((skel_rank_many_kernel<EPT, 32>), dim3(W, P), dim3(BLOCK), 0, st, dargs);
        else if (W <= 64) hipLaunchKernelGGL((skel_rank_many_kernel<EPT, 64>), dim3(W, P), dim3(BLOCK), 0, st, dargs);
        else hipLaunchKernelGGL((skel_rank_many_kernel<EPT, SKN_MAXW>), dim3(W, P), dim3(BLOCK), 0, st, dargs);
        return;
    }
    if (W <= 256) hipLaunchKernelGGL((skel_k2_many_kernel<4, 4>), dim3(SKK / 4, P), dim3(BLOCK), 0, st, dargs);
    else hipLaunchKernelGGL((skel_k2_many_kernel<4, 16>), dim3(SKK / 4, P), dim3(BLOCK), 0, st, dargs);
    hipLaunchKernelGGL((skel_rank_many_kernel<EPT, 0, true>), dim3(W, P), dim3(BLOCK), 0, st, dargs);     // (the radix-4 form, 20 KB of LDS: as the single-panel round)
}

extern "C" int pbwtamd_pass_advance_many(pbwtamd_engine **es, int P, const void *const *d_bitcols, int wpc, int ncols, int ncols_avail, unsigned opts) {
    if (P < 1 || !es || !d_bitcols) return fail("pbwtamd_pass_advance_many: no panels");
    pbwtamd_engine *e0 = es[0];
    HIPCHK(hipSetDevice(e0->device));
    bool fused = P > 1 && e0->skel && !e0->prow && e0->Wt <= 1024 && !(opts & PBWTAMD_OPT_SORTED) && !e0->sh && !e0->persist && !e0->sub_rounds;
    for (int p = 0; p < P; ++p) {
        pbwtamd_engine *e = es[p];
        if (!e->pass_open) return fail("pbwtamd_pass_advance_many: panel %d without pass_begin", p);
        if (e->M != e0->M || e->B != e0->B || e->stream != e0->stream || e->device != e0->device || e->k_cur != e0->k_cur || e->n_total != e0->n_total || e->ring != e0->ring)
            return fail("pbwtamd_pass_advance_many: panel %d differs from panel 0 in %s (same width, batch, stream, device and progress required)", p,
                        e->M != e0->M ? "width" : e->B != e0->B ? "batch" : e->stream != e0->stream ? "stream" : e->device != e0->device ? "device" : "progress");
        if (e->sh || e->persist || e->sub_rounds) fused = false;
    }
    if (wpc != e0->wpc) return fail("pbwtamd_pass_advance_many: wpc %d != engine wpc %d", wpc, e0->wpc);
    if (e0->k_cur + ncols > e0->n_total) return fail("pbwtamd_pass_advance_many: beyond n_total");
    if ((opts & (PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS | PBWTAMD_OPT_LONG_RECS)) && !(opts & PBWTAMD_OPT_WITH_D)) return fail("pbwtamd: the maxWithin sweep needs OPT_WITH_D");
    int done = 0;
    while (done < ncols) {
        const int nb = std::min(e0->B, ncols - done), left = ncols_avail - done, remaining = e0->n_total - e0->k_cur;
        if (!fused || nb % 8 || left < std::min(nb + 8, remaining)) {           // one panel after the other for this batch
            for (int p = 0; p < P; ++p) CHK(pbwtamd_pass_advance(es[p], (const uint32_t *)d_bitcols[p] + (size_t)done * wpc, wpc, nb, left, opts));
            done += nb;
            continue;
        }
        const int r = e0->ring, nr = nb / 8, W = e0->Wt;
        const bool two = skel_two_launch(e0);
        // per panel: what pbwtamd_pass_advance does in front of a skeleton batch
        for (int p = 0; p < P; ++p) {
            pbwtamd_engine *e = es[p];
            e->qs_bsum_sites[r] = 0;
            if (e->ev_used == e->ev.size()) { hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); e->ev.push_back({a, b}); }
            HIPCHK(hipEventRecord(e->ev[e->ev_used].first, e->stream));
            e->xT = e->xTr[r];
            CHK(skel_prepare(e, r, (const uint32_t *)d_bitcols[p] + (size_t)done * wpc, nb, left, false));
        }
        // the arguments of every round and panel, uploaded once (panel 0 owns the staging: two halves, reused two batches later)
        const size_t need = (size_t)(e0->B / 8 + 1) * (size_t)P;
        if (e0->margs_cap < need) {
            if (e0->margs) { HIPCHK(hipStreamSynchronize(e0->stream)); HIPCHK(dev_free(e0->margs)); HIPCHK(hipHostFree(e0->margs_host)); }
            HIPCHK(dev_alloc((void **)&e0->margs, 2 * need * sizeof(SkArgs)));
            HIPCHK(hipHostMalloc((void **)&e0->margs_host, 2 * need * sizeof(SkArgs), hipHostMallocDefault));
            e0->margs_cap = need;
            for (int i = 0; i < 2; ++i) if (!e0->evMargs[i]) HIPCHK(hipEventCreateWithFlags(&e0->evMargs[i], hipEventDisableTiming));
        }
        const int h = e0->margs_half; e0->margs_half ^= 1;
        HIPCHK(hipEventSynchronize(e0->evMargs[h]));
        SkArgs *host = e0->margs_host + (size_t)h * e0->margs_cap, *dev = e0->margs + (size_t)h * e0->margs_cap;
        for (int s8 = 0; s8 < nr; ++s8)
            for (int p = 0; p < P; ++p) host[(size_t)s8 * P + p] = skel_round_args(es[p], r, (const uint32_t *)d_bitcols[p] + (size_t)done * wpc, false, nb, left, s8, true);
        HIPCHK(hipMemcpyAsync(dev, host, (size_t)nr * P * sizeof(SkArgs), hipMemcpyHostToDevice, e0->stream));
        HIPCHK(hipEventRecord(e0->evMargs[h], e0->stream));
        auto rounds = [&](int s_from, int s_to) -> int {
            for (int s8 = s_from; s8 < s_to; ++s8) {
                const SkArgs *da = dev + (size_t)s8 * P;
                if (e0->skEPT == 1) launch_round_many<1>(e0, da, P, W, two); else if (e0->skEPT == 2) launch_round_many<2>(e0, da, P, W, two); else launch_round_many<4>(e0, da, P, W, two);
                if (e0->thr_rounds > 0 && (s8 + 1) % e0->thr_rounds == 0) {
                    HIPCHK(hipEventRecord(e0->tev[e0->tev_n % 16], e0->stream));
                    if (e0->tev_n >= e0->thr_depth) HIPCHK(hipEventSynchronize(e0->tev[(e0->tev_n - e0->thr_depth) % 16]));
                    ++e0->tev_n;
                }
            }
            HIPCHK(hipGetLastError());
            return 0;
        };
        CHK(rounds(0, nr - 1));
        for (int p = 0; p < P; ++p) CHK(flush_pending(es[p]));                    // the previous batch's consumers, beside this batch's chain
        for (int p = 0; p < P; ++p) if (es[p]->consRecorded[r ^ 1]) HIPCHK(hipStreamWaitEvent(e0->stream, es[p]->evCons[r ^ 1], 0));
        CHK(rounds(nr - 1, nr));                                                   // scatters into slot 0 (and the key row) of every panel's other ring
        for (int p = 0; p < P; ++p) {
            pbwtamd_engine *e = es[p];
            e->keys_ready[r ^ 1] = host[(size_t)(nr - 1) * P + p].has_next != 0;
            e->prepared = false;
            HIPCHK(hipEventRecord(e->ev[e->ev_used].second, e->stream));
            HIPCHK(hipEventRecord(e->evChain[r], e->stream)); e->chainRecorded[r] = true;
            ++e->ev_used; e->launches += (p == 0) ? (long long)(two ? 2 : 3) * nr : 0; e->sites_done += nb;
            e->pend.valid = true; e->pend.ring = r; e->pend.kbase = e->k_cur; e->pend.nb = nb; e->pend.opts = opts; e->pend.skel = true; e->pend.sharded = false;
            e->pend.early = false; e->pend.flushed = 0; e->pend.cols = (const uint32_t *)d_bitcols[p] + (size_t)done * wpc;
            e->ring_skel[r] = true;
            e->ring = r ^ 1; e->k_cur += nb;
        }
        done += nb;
    }
    return 0;
}

extern "C" int pbwtamd_pass_end(pbwtamd_engine *e, unsigned opts) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->pass_open) return fail("pbwtamd_pass_end without pass_begin");
    if (e->k_cur != e->n_total) return fail("pbwtamd_pass_end: at site %d of %d", e->k_cur, e->n_total);
    const bool with_d = opts & PBWTAMD_OPT_WITH_D;
    CHK(flush_pending(e));
    if (e->sh) CHK(shard_make_full(e));                    // every rank ends with the complete final state
    HIPCHK(hipStreamSynchronize(e->stream));               // the final state sits in slot 0 of e->ring
    const int *A = ringA(e, e->ring), *D = ringD(e, e->ring);
    if (e->sh) {                                           // the closing site belongs to the last rank; nobody leaves before every pull is done
        shard_xbar(e, e->stream, 1, 3, ++e->sh->e2);
        if (e->sh->rank != e->sh->world - 1) { e->pass_open = false; return pbwtamd_sync(e); }
    }
    if (opts & PBWTAMD_OPT_CHECKSUM) {
        unsigned long long *ca = e->csum + (e->k_cur - e->k0), *cd = ca + e->csum_sites, *cy = cd + e->csum_sites;
        dim3 grid(std::min(64, (e->M + BLOCK) / BLOCK), 1);
        hipLaunchKernelGGL(checksum_kernel, grid, dim3(BLOCK), 0, e->s2, A, D, e->strideA, e->strideD, e->M, with_d ? 1 : 0, ca, cd, cy, 0);
        HIPCHK(hipGetLastError());
    }
    if (opts & (PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_WITHIN_RECS)) CHK(run_within(e, e->s2, A, D, e->n_total, 1, 0, opts));
    if (opts & PBWTAMD_OPT_LONG_RECS) CHK(run_long(e, e->s2, A, D, e->ystale, e->n_total, 1, 0));
    e->pass_open = false;
    return pbwtamd_sync(e);
}

// close a pass before the panel's last site: the consumers of every batch advanced so far complete, no k == N sweep.
// For callers that own a block of sites only (site-block sharding across GPUs, pbwt_amd/siteblock.py).
extern "C" int pbwtamd_pass_stop(pbwtamd_engine *e) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->pass_open) return fail("pbwtamd_pass_stop without pass_begin");
    CHK(flush_pending(e));
    if (e->sh) { CHK(shard_make_full(e)); shard_xbar(e, e->stream, 1, 3, ++e->sh->e2); }
    e->pass_open = false;
    return pbwtamd_sync(e);
}

// restart a pass from a checkpoint (a_k, d_k) — the state pbwtCheckPoint / a cursor dump holds (pbwtIO.c:158-168 keeps a;
// d is what ForwardsAD needs in addition): pbwtamd_pass_begin at site k0 with the order, then the divergences
extern "C" int pbwtamd_pass_set_d(pbwtamd_engine *e, const int32_t *d) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->pass_open || e->k_cur != e->k0) return fail("pbwtamd_pass_set_d: only right after pbwtamd_pass_begin");
    if (d[0] != e->k0 + 1 || d[e->M] != e->k0 + 1) return fail("pbwtamd_pass_set_d: d[0] and d[M] must be the sentinels k0+1 = %d", e->k0 + 1);
    HIPCHK(h2d_async(ringD(e, e->ring), d, sizeof(int) * ((size_t)e->M + 1), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int pbwtamd_get_state(pbwtamd_engine *e, int32_t *a, int32_t *d) {
    HIPCHK(hipSetDevice(e->device));
    // position-sharded engine: between pass_begin and pass_end / pass_stop the cursor lives in the ranks' skeleton rings, slot 0 of the full ring is
    // stale, and completing it is a collective step (every rank pulls from every rank behind a barrier) — refuse instead of returning a stale state
    if (e->sh && e->sh->world > 1 && e->pass_open && !e->sh->full_state)
        return fail("pbwtamd_get_state: on a position-sharded engine the cursor is complete only after pbwtamd_pass_end / pbwtamd_pass_stop (mid-pass it is spread over the ranks)");
    CHK(flush_pending(e));
    HIPCHK(hipStreamSynchronize(e->s2));                   // ycols scratch is shared with pack3
    // slot 0 of the current ring holds the cursor; strip the allele tags through the ycols scratch
    const int *A = ringA(e, e->ring), *D = ringD(e, e->ring);
    int *tmp = (int *)e->ycols;
    if ((size_t)e->M * sizeof(int) > ((size_t)e->B + 1) * e->wpc64 * sizeof(unsigned long long)) {
        std::vector<int> h(e->M);
        HIPCHK(hipMemcpyAsync(h.data(), A, sizeof(int) * (size_t)e->M, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (int i = 0; i < e->M; ++i) a[i] = h[i] & AMASK;
    } else {
        hipLaunchKernelGGL(untag_kernel, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, A, tmp, e->M);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(a, tmp, sizeof(int) * (size_t)e->M, hipMemcpyDeviceToHost, e->stream));
    }
    if (d) HIPCHK(hipMemcpyAsync(d, D, sizeof(int) * ((size_t)e->M + 1), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int pbwtamd_get_hist(pbwtamd_engine *e, int64_t *hist, int histlen) {
    HIPCHK(hipSetDevice(e->device));
    CHK(flush_pending(e));
    const int n = std::min(histlen, e->histlen);
    memset(hist, 0, sizeof(int64_t) * (size_t)histlen);
    HIPCHK(hipStreamSynchronize(e->stream));               // pass_end's k == N sweep may have run on either stream
    hipLaunchKernelGGL(hist_fold_kernel, dim3((HIST_LBINS + 255) / 256), dim3(256), 0, e->s2, e->hist, e->hist_rep, e->histlen);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hist, e->hist, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, e->s2));
    HIPCHK(hipStreamSynchronize(e->s2));
    return 0;
}

extern "C" int pbwtamd_get_checksums(pbwtamd_engine *e, int k_first, int n, uint64_t *ca, uint64_t *cd, uint64_t *cy) {
    HIPCHK(hipSetDevice(e->device));
    CHK(flush_pending(e));
    const int off = k_first - e->k0;
    if (off < 0 || off + n > e->csum_sites) return fail("pbwtamd_get_checksums: range outside the pass");
    if (ca) HIPCHK(hipMemcpyAsync(ca, e->csum + off, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, e->s2));
    if (cd) HIPCHK(hipMemcpyAsync(cd, e->csum + e->csum_sites + off, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, e->s2));
    if (cy) HIPCHK(hipMemcpyAsync(cy, e->csum + 2 * (size_t)e->csum_sites + off, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, e->s2));
    HIPCHK(hipStreamSynchronize(e->s2));
    return 0;
}

extern "C" int pbwtamd_get_chain_timing(pbwtamd_engine *e, double *ms_total, int64_t *launches) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    double tot = 0;
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->ev[i].first, e->ev[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = e->launches;
    return 0;
}

extern "C" int pbwtamd_get_chain_sites(pbwtamd_engine *e, int64_t *sites) { *sites = e->sites_done; return 0; }

extern "C" int pbwtamd_get_phase_profile(pbwtamd_engine *e, int64_t *out, int ntiles) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->prof) return -fail("pbwtamd_get_phase_profile: engine created without PBWTAMD_PROFILE=1");
    HIPCHK(hipStreamSynchronize(e->stream));
    const int n = std::min(ntiles, e->W);
    HIPCHK(hipMemcpy(out, e->prof, (size_t)n * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return n;
}

extern "C" int pbwtamd_synth_device(pbwtamd_engine *e, void *d_bitcols, int k0, int ncols, uint64_t seed, int kind) {
    HIPCHK(hipSetDevice(e->device));
    int done = 0;
    while (done < ncols) {
        const int nb = std::min(ncols - done, 32768 * SYNTH_CPB);
        dim3 grid((e->wpc * 32 + BLOCK - 1) / BLOCK, (nb + SYNTH_CPB - 1) / SYNTH_CPB);    // every word of a column: wpc is padded beyond ceil(M / 32)
        hipLaunchKernelGGL(synth_kernel, grid, dim3(BLOCK), 0, e->stream, (uint32_t *)d_bitcols + (size_t)done * e->wpc, e->M, k0 + done, nb, e->wpc, seed, kind);
        HIPCHK(hipGetLastError());
        done += nb;
    }
    return 0;
}

// ------------------------------------------------------------------------------------ host-buffer API
extern "C" int pbwtamd_get_packed(pbwtamd_engine *e, uint8_t **yz_out, int64_t *nz_out) {
    HIPCHK(hipSetDevice(e->device));
    CHK(pbwtamd_sync(e));
    unsigned long long nz = 0;
    HIPCHK(hipMemcpy(&nz, e->scal + 1, sizeof nz, hipMemcpyDeviceToHost));
    uint8_t *buf = (uint8_t *)malloc(nz ? nz : 1);
    if (!buf) return fail("pbwtamd_get_packed: out of host memory for %llu bytes", nz);
    if (nz) HIPCHK(hipMemcpy(buf, e->yz, nz, hipMemcpyDeviceToHost));
    *yz_out = buf; *nz_out = (int64_t)nz;
    return 0;
}

extern "C" int pbwtamd_build(pbwtamd_engine *e, const uint32_t *bitcols, int wpc, int N, int with_d,
                             const int32_t *aFstart, uint8_t **yz_out, int64_t *nz_out, int32_t *aFend, int32_t *dFend) {
    HIPCHK(hipSetDevice(e->device));
    if (wpc < (e->M + 31) / 32) return fail("pbwtamd_build: wpc %d too small for M %d", wpc, e->M);
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    const unsigned opts = (with_d ? PBWTAMD_OPT_WITH_D : 0u) | (yz_out ? PBWTAMD_OPT_PACK3 : 0u);
    // pin the caller's columns for the duration of the build: the per-batch copies then run as DMA at link speed beside the
    // chain instead of through the runtime's bounce buffers (falls back to pageable copies if registration is refused)
    static const bool no_pin = tune_env("PBWTAMD_NO_PIN") != nullptr;
    const size_t in_bytes = (size_t)N * wpc * sizeof(uint32_t);
    const bool pinned = !no_pin && in_bytes >= (1u << 20) && hipHostRegister((void *)bitcols, in_bytes, hipHostRegisterDefault) == hipSuccess;
    if (!pinned) (void)hipGetLastError();
    // on any exit — errors included — the async copies out of the caller's buffer are drained before it is unregistered
    struct Unpin { const void *p; bool on; hipStream_t st; ~Unpin() { if (on) { (void)hipStreamSynchronize(st); (void)hipHostUnregister((void *)p); } } } unpin{bitcols, pinned, e->stream};
    int done = 0, half = 0;
    while (done < N) {
        const int nb = std::min(e->B, N - done);
        const int navail = std::min(nb + 8, N - done);       // look-ahead: the skeleton chain's radix step spans 8 sites
        uint32_t *stage = e->cols_stage + (size_t)half * (e->B + 8) * e->wpc;
        // this half was read by the chain two batches ago (same ring): wait for that chain, not for the one in flight
        if (e->chainRecorded[e->ring]) HIPCHK(hipEventSynchronize(e->evChain[e->ring]));
        if (wpc == e->wpc)
            HIPCHK(h2d_async(stage, bitcols + (size_t)done * wpc, (size_t)navail * wpc * sizeof(uint32_t), e->stream));
        else {
            HIPCHK(hipMemsetAsync(stage, 0, (size_t)navail * e->wpc * sizeof(uint32_t), e->stream));
            HIPCHK(hipMemcpy2DAsync(stage, (size_t)e->wpc * 4, bitcols + (size_t)done * wpc, (size_t)wpc * 4,
                                    (size_t)std::min(wpc, e->wpc) * 4, (size_t)navail, hipMemcpyHostToDevice, e->stream));
        }
        CHK(pbwtamd_pass_advance(e, stage, e->wpc, nb, navail, opts));
        half ^= 1;
        done += nb;
    }
    CHK(pbwtamd_pass_end(e, opts));
    if (aFend) CHK(pbwtamd_get_state(e, aFend, with_d ? dFend : nullptr));
    if (yz_out) CHK(pbwtamd_get_packed(e, yz_out, nz_out));
    return 0;
}

// decode state for packed panels on the device
struct Packed {
    uint8_t *z = nullptr; long long *colStart = nullptr; unsigned long long *blockSum = nullptr;
    ~Packed() { if (z) (void)dev_free(z); if (colStart) (void)dev_free(colStart); if (blockSum) (void)dev_free(blockSum); }
};

static int packed_upload(pbwtamd_engine *e, hipStream_t st, int M, const uint8_t *yz, int64_t nz, int N, Packed &pk) {
    if (nz <= 0 && N > 0) return fail("pbwtamd: empty packed panel for N=%d", N);
    HIPCHK(dev_alloc((void **)&pk.z, (size_t)std::max<int64_t>(nz, 1)));
    HIPCHK(h2d_async(pk.z, yz, (size_t)nz, st));
    const size_t nblk = ((size_t)nz + DEC_CHUNK - 1) / DEC_CHUNK;
    HIPCHK(dev_alloc((void **)&pk.blockSum, (nblk + 1) * sizeof(unsigned long long)));
    HIPCHK(dev_alloc((void **)&pk.colStart, ((size_t)N + 2) * sizeof(long long)));
    HIPCHK(hipMemsetAsync(pk.colStart, 0xff, ((size_t)N + 2) * sizeof(long long), st));
    if (nblk) {
        hipLaunchKernelGGL(dec_sum_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, (const uint8_t *)pk.z, (size_t)nz, pk.blockSum);
        hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, pk.blockSum, nblk, pk.blockSum + nblk, 0ULL);
        hipLaunchKernelGGL(dec_colstart_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, (const uint8_t *)pk.z, (size_t)nz, (const unsigned long long *)pk.blockSum, M, (long long)N, pk.colStart);
        HIPCHK(hipGetLastError());
    }
    unsigned long long total = 0;
    if (nblk) HIPCHK(hipMemcpyAsync(&total, pk.blockSum + nblk, sizeof total, hipMemcpyDeviceToHost, st));
    const long long end = nz;
    HIPCHK(h2d_async(pk.colStart + N, &end, sizeof end, st));
    HIPCHK(hipStreamSynchronize(st));
    if (total != (unsigned long long)M * (unsigned long long)N)
        return fail("pbwtamd: packed panel decodes to %llu alleles, expected M*N = %llu", total, (unsigned long long)M * (unsigned long long)N);
    if (N > 0) {                                           // every column boundary found, in order, <= M bytes apart: before any expand
        int *bad = nullptr, hbad = 0;
        DevBufs tmp;
        CHK(tmp.alloc(&bad, 1));
        HIPCHK(hipMemsetAsync(bad, 0, sizeof(int), st));
        hipLaunchKernelGGL(dec_validate_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const long long *)pk.colStart, (long long)N, (long long)nz, M, bad);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hbad) return fail("pbwtamd: malformed packed panel (a run straddles a column boundary, or a column of more than M bytes)");
    }
    (void)e;
    return 0;
}

// expand columns [c0, c0+nc) of a packed panel into ycols (wpc64 words per column)
static int packed_expand(pbwtamd_engine *e, hipStream_t st, const Packed &pk, int M, long long c0, int nc, unsigned long long *ycols, int wpc64) {
    HIPCHK(hipMemsetAsync(ycols, 0, (size_t)nc * wpc64 * sizeof(unsigned long long), st));
    if (nc) hipLaunchKernelGGL(dec_expand_kernel, dim3(nc), dim3(BLOCK), 0, st, (const uint8_t *)pk.z, (const long long *)pk.colStart, c0, M, ycols, wpc64, e->ctl + 2);
    HIPCHK(hipGetLastError());
    return 0;
}

// drive a read-side pass over a packed panel; per batch decode -> ycols -> chain (+consumers)
static int get_state_y(pbwtamd_engine *e, uint8_t *y) {
    CHK(flush_pending(e));
    HIPCHK(hipStreamSynchronize(e->s2));
    unsigned char *tmp = (unsigned char *)e->ycols;
    hipLaunchKernelGGL(tags_to_bytes_kernel, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, (const int *)ringA(e, e->ring), tmp, e->M);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y, tmp, (size_t)e->M, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// [rec_lo, rec_hi): the sites whose states the consumers in `opts` see (default: all of 0..N); the chain runs over every site
static int sweep_packed(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, unsigned opts,
                        const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump, uint8_t *y_dump = nullptr,
                        int rec_lo = 0, int rec_hi = 0x7fffffff) {
    Packed pk;
    CHK(packed_upload(e, e->stream, e->M, yz, nz, N, pk));
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    opts |= PBWTAMD_OPT_SORTED | PBWTAMD_OPT_WITH_D;
    std::vector<int> tmp(e->M);
    auto dump_at = [&](int k) -> int {
        for (int q = 0; q < ndump; ++q) if (dump_sites[q] == k) {
            CHK(pbwtamd_get_state(e, a_dump + (size_t)q * e->M, d_dump ? d_dump + (size_t)q * (e->M + 1) : nullptr));
            if (y_dump) CHK(get_state_y(e, y_dump + (size_t)q * e->M));
        }
        return 0;
    };
    int done = 0;
    bool any_dump = ndump > 0;
    while (done < N) {
        int nb = std::min(e->B, N - done);
        if (any_dump) {                                      // stop at the next dump site
            int nxt = N + 1;
            for (int q = 0; q < ndump; ++q) if (dump_sites[q] > done && dump_sites[q] < nxt) nxt = dump_sites[q];
            nb = std::min(nb, nxt - done);
        }
        if (done < rec_lo) nb = std::min(nb, rec_lo - done);   // batches do not straddle the window's ends
        else if (done < rec_hi) nb = std::min(nb, rec_hi - done);
        const bool in_window = done >= rec_lo && done < rec_hi;
        const int navail = std::min(nb + 1, N - done);
        // decode straight into the column staging buffer (ycols is scratch for pack3/get_state)
        CHK(packed_expand(e, e->stream, pk, e->M, done, navail, (unsigned long long *)e->cols_stage, e->wpc64));
        if (any_dump) {
            CHK(ensure_prepared(e, e->cols_stage, true, true, false, navail));   // tags of the first site exist before it is dumped
            CHK(dump_at(done));
        }
        CHK(pbwtamd_pass_advance(e, e->cols_stage, e->wpc, nb, navail, in_window ? opts : (PBWTAMD_OPT_SORTED | PBWTAMD_OPT_WITH_D)));
        done += nb;
    }
    if (any_dump) CHK(dump_at(N));
    CHK(pbwtamd_pass_end(e, (N >= rec_lo && N < rec_hi) ? opts : (PBWTAMD_OPT_SORTED | PBWTAMD_OPT_WITH_D)));
    return 0;
}

extern "C" int pbwtamd_sweep_AD(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                                uint64_t *csum_a, uint64_t *csum_d, uint64_t *csum_y,
                                const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump, uint8_t *y_dump) {
    HIPCHK(hipSetDevice(e->device));
    const unsigned opts = (csum_a || csum_d || csum_y) ? PBWTAMD_OPT_CHECKSUM : 0u;
    CHK(sweep_packed(e, yz, nz, N, aFstart, opts, dump_sites, ndump, a_dump, d_dump, y_dump));
    if (opts) CHK(pbwtamd_get_checksums(e, 0, N + 1, csum_a, csum_d, csum_y));
    return 0;
}

extern "C" int pbwtamd_max_within(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                                  pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out,
                                  int64_t *hist, int histlen) {
    HIPCHK(hipSetDevice(e->device));
    const int sinks = (report ? 1 : 0) + (recs_out ? 1 : 0) + (hist ? 1 : 0);
    if (sinks != 1) return fail("pbwtamd_max_within: exactly one of report / recs_out / hist must be given");
    if (e->M < 2) return fail("pbwtamd_max_within: needs at least 2 haplotypes (the reference reads y[-1] for M = 1)");
    if (hist && histlen < N + 1) return fail("pbwtamd_max_within: histlen %d < N+1", histlen);
    std::vector<pbwtamd_match> recs;
    e->rec_sink = &recs; e->rec_cb = report;
    const unsigned opts = hist ? PBWTAMD_OPT_WITHIN_HIST : PBWTAMD_OPT_WITHIN_RECS;
    const int rc = sweep_packed(e, yz, nz, N, aFstart, opts, nullptr, 0, nullptr, nullptr);
    e->rec_sink = nullptr; e->rec_cb = nullptr;
    if (rc) return rc;
    if (hist) CHK(pbwtamd_get_hist(e, hist, histlen));
    if (recs_out) {
        pbwtamd_match *buf = (pbwtamd_match *)malloc(std::max<size_t>(1, recs.size()) * sizeof(pbwtamd_match));
        if (!buf) return fail("pbwtamd_max_within: out of host memory");
        if (!recs.empty()) memcpy(buf, recs.data(), recs.size() * sizeof(pbwtamd_match));
        *recs_out = buf; *nrecs_out = (int64_t)recs.size();
    }
    return 0;
}

// PbwtCursor view (pbwt.h:74-87) of the read-side cursor before site k: what pbwtCursorCreate(p,TRUE,TRUE) followed by k calls of
// pbwtCursorForwardsReadAD (pbwtCore.c:420-445,543-557) leave in u->a, u->d, u->y, u->c, plus u->u as pbwtCursorCalculateU would
// fill it, and the cursor's byte offsets into yz (u->nBlockStart, u->n; isBlockEnd = k < N).  y, c, u come from the packed column
// itself (y_k in sorted order IS column k of yz); at k == N they are the stale column N-1, as in the reference.
extern "C" int pbwtamd_cursor_at(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, int k,
                                 int32_t *a, int32_t *d, uint8_t *y, int32_t *c, int32_t *u, int64_t *nBlockStart, int64_t *n) {
    HIPCHK(hipSetDevice(e->device));
    if (k < 0 || k > N) return fail("pbwtamd_cursor_at: site %d outside 0..%d", k, N);
    Packed pk;
    CHK(packed_upload(e, e->stream, e->M, yz, nz, N, pk));
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    for (int done = 0; done < k;) {
        const int nb = std::min(e->B, k - done), navail = std::min(nb + 1, N - done);
        CHK(packed_expand(e, e->stream, pk, e->M, done, navail, (unsigned long long *)e->cols_stage, e->wpc64));
        CHK(pbwtamd_pass_advance(e, e->cols_stage, e->wpc, nb, navail, PBWTAMD_OPT_SORTED | PBWTAMD_OPT_WITH_D));
        done += nb;
    }
    if (a) CHK(pbwtamd_get_state(e, a, d));
    e->pass_open = false;
    CHK(pbwtamd_sync(e));
    const int ky = std::min(k, N - 1);
    if (ky >= 0) {
        DevBufs bufs;
        unsigned char *dy; int *du, *rd;
        CHK(bufs.alloc(&dy, (size_t)e->M)); CHK(bufs.alloc(&du, (size_t)e->M + 1)); CHK(bufs.alloc(&rd, (size_t)e->wpc64 + 1));
        CHK(packed_expand(e, e->stream, pk, e->M, ky, 1, e->ycols, e->wpc64));
        hipLaunchKernelGGL(qs_rankdir_kernel, dim3(1), dim3(BLOCK), 0, e->stream, (const unsigned long long *)e->ycols, e->wpc64, e->M, rd);
        hipLaunchKernelGGL(cursor_y_u_kernel, dim3((e->M + 256) / 256), dim3(256), 0, e->stream, (const unsigned long long *)e->ycols, (const int *)rd, e->M, dy, du);
        HIPCHK(hipGetLastError());
        if (y) HIPCHK(hipMemcpyAsync(y, dy, (size_t)e->M, hipMemcpyDeviceToHost, e->stream));
        if (u) HIPCHK(hipMemcpyAsync(u, du, sizeof(int) * ((size_t)e->M + 1), hipMemcpyDeviceToHost, e->stream));
        if (c) HIPCHK(hipMemcpyAsync(c, rd + e->wpc64, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        long long cs[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cs, pk.colStart + ky, sizeof cs, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (nBlockStart) *nBlockStart = cs[0];
        if (n) *n = (k < N) ? cs[1] : nz;
        CHK(pbwtamd_sync(e));                               // device error flag of the expand
    } else {                                                // empty panel: pbwtCursorCreate leaves n = 0, y unset (pbwtCore.c:436-439)
        if (y) memset(y, 0, (size_t)e->M);
        if (u) memset(u, 0, sizeof(int) * ((size_t)e->M + 1));
        if (c) *c = 0;
        if (nBlockStart) *nBlockStart = 0;
        if (n) *n = 0;
    }
    return 0;
}

// matchMaximalWithin with the reports restricted to the sites k_lo <= k < k_hi (k_hi <= N + 1; k == N is the final
// all-positions report of pbwtMatch.c:126 `k < p->N`): what a caller's report() sees if it ignores every other `end`
extern "C" int pbwtamd_max_within_range(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                                        int k_lo, int k_hi, pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out) {
    HIPCHK(hipSetDevice(e->device));
    if ((report ? 1 : 0) + (recs_out ? 1 : 0) != 1) return fail("pbwtamd_max_within_range: exactly one of report / recs_out must be given");
    if (e->M < 2) return fail("pbwtamd_max_within_range: needs at least 2 haplotypes");
    if (k_lo < 0 || k_hi > N + 1 || k_lo > k_hi) return fail("pbwtamd_max_within_range: window [%d, %d) outside 0..%d", k_lo, k_hi, N + 1);
    std::vector<pbwtamd_match> recs;
    e->rec_sink = &recs; e->rec_cb = report;
    const int rc = sweep_packed(e, yz, nz, N, aFstart, PBWTAMD_OPT_WITHIN_RECS, nullptr, 0, nullptr, nullptr, nullptr, k_lo, k_hi);
    e->rec_sink = nullptr; e->rec_cb = nullptr;
    if (rc) return rc;
    if (recs_out) {
        pbwtamd_match *buf = (pbwtamd_match *)malloc(std::max<size_t>(1, recs.size()) * sizeof(pbwtamd_match));
        if (!buf) return fail("pbwtamd_max_within_range: out of host memory");
        if (!recs.empty()) memcpy(buf, recs.data(), recs.size() * sizeof(pbwtamd_match));
        *recs_out = buf; *nrecs_out = (int64_t)recs.size();
    }
    return 0;
}

// -haps (pbwtWriteHaplotypes, pbwtIO.c:839-857): the panel's alleles in original haplotype order,
// out[k*M + h] = 0/1, from a forward A-only sweep of the packed panel
extern "C" int pbwtamd_haplotypes(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, uint8_t *out) {
    HIPCHK(hipSetDevice(e->device));
    Packed pk;
    CHK(packed_upload(e, e->stream, e->M, yz, nz, N, pk));
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    DevBufs bufs;
    unsigned char *dout;
    CHK(bufs.alloc(&dout, (size_t)e->B * e->M));
    for (int done = 0; done < N;) {
        const int nb = std::min(e->B, N - done);
        const int navail = std::min(nb + 1, N - done);
        CHK(packed_expand(e, e->stream, pk, e->M, done, navail, (unsigned long long *)e->cols_stage, e->wpc64));
        CHK(pbwtamd_pass_advance(e, e->cols_stage, e->wpc, nb, navail, PBWTAMD_OPT_SORTED | OPT_INTERNAL_KEEP_STATES));
        CHK(pbwtamd_sync(e));                              // every state of the batch is in the ring (incl. the fill of the skeleton path)
        const int *A = ringA(e, e->ring ^ 1);
        dim3 grid(std::min(64, (e->M + BLOCK - 1) / BLOCK), nb);
        hipLaunchKernelGGL(unsort_alleles_kernel, grid, dim3(BLOCK), 0, e->stream, A, e->strideA, e->M, dout);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out + (size_t)done * e->M, dout, (size_t)nb * e->M, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        done += nb;
    }
    return pbwtamd_pass_end(e, PBWTAMD_OPT_SORTED);
}

// Panel transforms on the device: decode the packed panel with a forward sweep (original-order alleles of every batch of sites),
// gather the selected haplotypes of the selected sites into the new panel's bit columns (kept in HBM), and run the build chain
// over them.  One entry point covers pbwtBuildReverse (site_order = N-1 .. 0, start order = the forward panel's final order),
// pbwtSubSample (hap_select), pbwtSubRange / pbwtSelectSites / pbwtRemoveSites (site_order = the kept sites, increasing).
extern "C" int pbwtamd_regather(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                                const int32_t *site_order, int n_out, const int32_t *hap_select, int M_out, const int32_t *aStart_out,
                                uint8_t **yz_out, int64_t *nz_out, int32_t *aFend_out, int32_t *aFend_fwd) {
    HIPCHK(hipSetDevice(e->device));
    const int M = e->M;
    if (!hap_select) M_out = M;
    if (!site_order) n_out = N;
    if (M_out < 1 || n_out < 0) return fail("pbwtamd_regather: M_out %d, n_out %d", M_out, n_out);
    std::vector<int> inv((size_t)N + 1, -1);                // input site -> output column
    int last_needed = -1;
    for (int j = 0; j < n_out; ++j) {
        const int sIn = site_order ? site_order[j] : j;
        if (sIn < 0 || sIn >= N) return fail("pbwtamd_regather: site_order[%d] = %d outside 0..%d", j, sIn, N - 1);
        if (inv[(size_t)sIn] >= 0) return fail("pbwtamd_regather: input site %d selected twice", sIn);
        inv[(size_t)sIn] = j; last_needed = std::max(last_needed, sIn);
    }
    if (hap_select) for (int h = 0; h < M_out; ++h) if (hap_select[h] < 0 || hap_select[h] >= M) return fail("pbwtamd_regather: hap_select[%d] = %d outside 0..%d", h, hap_select[h], M - 1);
    const int wpc_out = wpc_for(M_out), wpc64_out = wpc_out / 2;
    const int n_sweep = aFend_fwd ? N : last_needed + 1;     // the forward order at the end needs the whole sweep
    DevBufs bufs;
    Packed pk;
    unsigned long long *cols_out; int *d_inv, *d_sel = nullptr; unsigned char *dout;
    CHK(bufs.alloc(&cols_out, (size_t)std::max(n_out, 1) * wpc64_out + 2 * (size_t)wpc64_out));
    CHK(bufs.alloc(&d_inv, (size_t)N + 1));
    CHK(bufs.alloc(&dout, (size_t)e->B * M));
    HIPCHK(h2d_async(d_inv, inv.data(), sizeof(int) * ((size_t)N + 1), e->stream));
    if (hap_select) { CHK(bufs.alloc(&d_sel, (size_t)M_out)); HIPCHK(h2d_async(d_sel, hap_select, sizeof(int) * (size_t)M_out, e->stream)); }
    HIPCHK(hipMemsetAsync(cols_out, 0, ((size_t)std::max(n_out, 1) * wpc64_out + 2 * (size_t)wpc64_out) * sizeof(unsigned long long), e->stream));
    CHK(packed_upload(e, e->stream, M, yz, nz, N, pk));
    // ---- phase A: forward sweep (A only), alleles of each batch back in original order, gathered into the new columns
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    for (int done = 0; done < n_sweep;) {
        const int nb = std::min(e->B, n_sweep - done), navail = std::min(nb + 1, N - done);
        CHK(packed_expand(e, e->stream, pk, M, done, navail, (unsigned long long *)e->cols_stage, e->wpc64));
        CHK(pbwtamd_pass_advance(e, e->cols_stage, e->wpc, nb, navail, PBWTAMD_OPT_SORTED | OPT_INTERNAL_KEEP_STATES));
        CHK(pbwtamd_sync(e));                              // every state of the batch is in the ring (incl. the fill of the skeleton path)
        const int *A = ringA(e, e->ring ^ 1);
        dim3 gu(std::min(64, (M + BLOCK - 1) / BLOCK), nb);
        hipLaunchKernelGGL(unsort_alleles_kernel, gu, dim3(BLOCK), 0, e->stream, A, e->strideA, M, dout);
        dim3 gg(std::min(64, (wpc64_out + WAVES - 1) / WAVES), nb);
        hipLaunchKernelGGL(regather_kernel, gg, dim3(BLOCK), 0, e->stream, (const unsigned char *)dout, M, (const int *)(d_inv + done), (const int *)d_sel, M_out, cols_out, wpc64_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(e->stream));
        done += nb;
    }
    if (aFend_fwd) CHK(pbwtamd_get_state(e, aFend_fwd, nullptr));
    e->pass_open = false;
    CHK(pbwtamd_sync(e));
    // ---- phase B: the build chain over the new columns (already resident)
    pbwtamd_engine *eb = e;
    struct EngGuard { pbwtamd_engine *p = nullptr; ~EngGuard() { if (p) pbwtamd_engine_destroy(p); } } guard;
    if (M_out != M) { CHK(pbwtamd_engine_create(&eb, e->device, M_out, e->B, nullptr)); guard.p = eb; }
    CHK(pbwtamd_pass_begin(eb, aStart_out, 0, n_out));
    const unsigned opts = yz_out ? PBWTAMD_OPT_PACK3 : 0u;
    if (n_out) CHK(pbwtamd_pass_advance(eb, cols_out, wpc_out, n_out, n_out, opts));
    CHK(pbwtamd_pass_end(eb, opts));
    if (aFend_out) CHK(pbwtamd_get_state(eb, aFend_out, nullptr));
    if (yz_out) CHK(pbwtamd_get_packed(eb, yz_out, nz_out));
    return 0;
}

// pbwtWriteHaplotypes without the N x M host matrix: `sink` receives the alleles of consecutive sites, nsites rows of M bytes
// (0/1, original haplotype order) at a time, on the calling thread, in site order
extern "C" int pbwtamd_haplotypes_stream(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                                         void (*sink)(int k0, int nsites, const uint8_t *rows, void *ctx), void *ctx) {
    HIPCHK(hipSetDevice(e->device));
    Packed pk;
    CHK(packed_upload(e, e->stream, e->M, yz, nz, N, pk));
    CHK(pbwtamd_pass_begin(e, aFstart, 0, N));
    DevBufs bufs;
    unsigned char *dout;
    CHK(bufs.alloc(&dout, (size_t)e->B * e->M));
    std::vector<uint8_t> rows((size_t)e->B * e->M);
    for (int done = 0; done < N;) {
        const int nb = std::min(e->B, N - done), navail = std::min(nb + 1, N - done);
        CHK(packed_expand(e, e->stream, pk, e->M, done, navail, (unsigned long long *)e->cols_stage, e->wpc64));
        CHK(pbwtamd_pass_advance(e, e->cols_stage, e->wpc, nb, navail, PBWTAMD_OPT_SORTED | OPT_INTERNAL_KEEP_STATES));
        CHK(pbwtamd_sync(e));
        dim3 grid(std::min(64, (e->M + BLOCK - 1) / BLOCK), nb);
        hipLaunchKernelGGL(unsort_alleles_kernel, grid, dim3(BLOCK), 0, e->stream, (const int *)ringA(e, e->ring ^ 1), e->strideA, e->M, dout);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(rows.data(), dout, (size_t)nb * e->M, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        sink(done, nb, rows.data(), ctx);
        done += nb;
    }
    return pbwtamd_pass_end(e, PBWTAMD_OPT_SORTED);
}

// -longWithin L: matchLongWithin2 (pbwtMatch.c:85-113) over a packed panel
extern "C" int pbwtamd_long_within(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, int L,
                                   pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out) {
    HIPCHK(hipSetDevice(e->device));
    if ((report ? 1 : 0) + (recs_out ? 1 : 0) != 1) return fail("pbwtamd_long_within: exactly one of report / recs_out must be given");
    if (L < 0) return fail("L %d for longWithin must be >= 0", L);
    std::vector<pbwtamd_match> recs;
    e->rec_sink = &recs; e->rec_cb = report; e->longL = L;
    const int rc = sweep_packed(e, yz, nz, N, aFstart, PBWTAMD_OPT_LONG_RECS, nullptr, 0, nullptr, nullptr);
    e->rec_sink = nullptr; e->rec_cb = nullptr;
    if (rc) return rc;
    if (recs_out) {
        pbwtamd_match *buf = (pbwtamd_match *)malloc(std::max<size_t>(1, recs.size()) * sizeof(pbwtamd_match));
        if (!buf) return fail("pbwtamd_long_within: out of host memory");
        if (!recs.empty()) memcpy(buf, recs.data(), recs.size() * sizeof(pbwtamd_match));
        *recs_out = buf; *nrecs_out = (int64_t)recs.size();
    }
    return 0;
}

extern "C" int pbwtamd_pack3(pbwtamd_engine *e, const uint32_t *sorted_bitcols, int wpc, int N, uint8_t **yz_out, int64_t *nz_out) {
    HIPCHK(hipSetDevice(e->device));
    if (wpc != e->wpc) return fail("pbwtamd_pack3: wpc %d != engine wpc %d", wpc, e->wpc);
    std::vector<uint8_t> all;
    unsigned long long *off = nullptr;
    for (int done = 0; done < N; done += e->B) {
        const int nb = std::min(e->B, N - done);
        HIPCHK(h2d_async(e->ycols, sorted_bitcols + (size_t)done * wpc, (size_t)nb * wpc * 4, e->stream));
        launch_p3r_sizes(e->stream, nb, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes);
        hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, e->stream, e->colBytes, (size_t)nb, e->scal + 2, 0ULL);
        HIPCHK(hipGetLastError());
        unsigned long long tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, e->scal + 2, sizeof tot, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        CHK(ensure_yz(e, e->stream, (size_t)tot + 16));
        launch_p3r_emit(e->stream, nb, (const unsigned long long *)e->ycols, e->wpc64, e->M, e->p3regs, e->colBytes, e->yz);
        HIPCHK(hipGetLastError());
        const size_t old = all.size();
        all.resize(old + tot);
        if (tot) HIPCHK(hipMemcpyAsync(all.data() + old, e->yz, tot, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    (void)off;
    uint8_t *buf = (uint8_t *)malloc(std::max<size_t>(1, all.size()));
    if (!buf) return fail("pbwtamd_pack3: out of host memory");
    if (!all.empty()) memcpy(buf, all.data(), all.size());
    *yz_out = buf; *nz_out = (int64_t)all.size();
    return 0;
}

extern "C" int pbwtamd_unpack3(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, uint32_t *sorted_bitcols, int wpc) {
    HIPCHK(hipSetDevice(e->device));
    if (wpc != e->wpc) return fail("pbwtamd_unpack3: wpc %d != engine wpc %d", wpc, e->wpc);
    Packed pk;
    CHK(packed_upload(e, e->stream, e->M, yz, nz, N, pk));
    for (int done = 0; done < N; done += e->B) {
        const int nb = std::min(e->B, N - done);
        CHK(packed_expand(e, e->stream, pk, e->M, done, nb, e->ycols, e->wpc64));
        HIPCHK(hipMemcpyAsync(sorted_bitcols + (size_t)done * wpc, e->ycols, (size_t)nb * wpc * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return pbwtamd_sync(e);
}


extern "C" int pbwtamd_match_sweep_sparse(pbwtamd_engine *e, const uint8_t *pz, int64_t pnz, int N, const int32_t *pStart,
                                          int Mq, const uint8_t *qz, int64_t qnz, const int32_t *qStart, int nSparse,
                                          pbwtamd_report5_fn report, pbwtamd_match5 **recs_out, int64_t *nrecs_out,
                                          int64_t *n_nomatch, int64_t *tot_out);

// Query sharding across GPUs ("for matchDynamic shard queries": queries are independent given the panel state,
// pbwtMatch.c:376-414).  After this call the query sweeps of `e` report for the queries lo <= jj < hi only (original indices of
// the query panel); every record's `sparse` field then carries, above bit 0, the query's rank in the query panel's order at the
// record's site — (end, rank, isSparse) is the reference's emission order, so per-rank streams merge exactly.  lo < 0: all queries.
extern "C" int pbwtamd_set_query_range(pbwtamd_engine *e, int lo, int hi) {
    if (lo < 0) { e->q_lo = 0; e->q_hi = 0x7fffffff; e->q_part = false; return 0; }
    if (hi < lo) return fail("pbwtamd_set_query_range: [%d, %d)", lo, hi);
    e->q_lo = lo; e->q_hi = hi; e->q_part = true;
    return 0;
}

// matchSequencesSweep (pbwtMatch.c:363-443) = the sparse sweep without sparse cursors (same kernels: one wave per query)
static thread_local pbwtamd_report_fn g_report4 = nullptr;
static void report4_thunk(int ai, int bi, int start, int end, int) { g_report4(ai, bi, start, end); }

extern "C" int pbwtamd_match_sweep(pbwtamd_engine *e, const uint8_t *pz, int64_t pnz, int N, const int32_t *pStart,
                                   int Mq, const uint8_t *qz, int64_t qnz, const int32_t *qStart,
                                   pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out,
                                   int64_t *n_nomatch, int64_t *tot_out) {
    if ((report ? 1 : 0) + (recs_out ? 1 : 0) != 1) return fail("pbwtamd_match_sweep: exactly one of report / recs_out must be given");
    if (report) {
        g_report4 = report;
        const int rc = pbwtamd_match_sweep_sparse(e, pz, pnz, N, pStart, Mq, qz, qnz, qStart, 0, report4_thunk, nullptr, nullptr, n_nomatch, tot_out);
        g_report4 = nullptr;
        return rc;
    }
    pbwtamd_match5 *r5 = nullptr; int64_t n5 = 0;
    CHK(pbwtamd_match_sweep_sparse(e, pz, pnz, N, pStart, Mq, qz, qnz, qStart, 0, nullptr, &r5, &n5, n_nomatch, tot_out));
    // five fields to four in place: record i ends at byte 16 i + 16 <= 20 (i + 1), where record i + 1 of the source starts
    unsigned char *raw = (unsigned char *)r5;
    for (int64_t i = 0; i < n5; ++i) { pbwtamd_match5 v; memcpy(&v, raw + 20 * i, sizeof v); const pbwtamd_match m = {v.ai, v.bi, v.start, v.end}; memcpy(raw + 16 * i, &m, sizeof m); }
    static_assert(sizeof(pbwtamd_match5) == 20 && sizeof(pbwtamd_match) == 16, "record layouts");
    *recs_out = (pbwtamd_match *)raw; *nrecs_out = n5;
    return 0;
}

extern "C" int pbwtamd_get_nomatch_events(pbwtamd_engine *e, int32_t **events, int64_t *n) {
    const size_t cnt = e->nomatch_events.size();
    int32_t *buf = (int32_t *)malloc(std::max<size_t>(1, cnt) * sizeof(int32_t));
    if (!buf) return fail("pbwtamd_get_nomatch_events: out of host memory");
    if (cnt) memcpy(buf, e->nomatch_events.data(), cnt * sizeof(int32_t));
    *events = buf; *n = (int64_t)(cnt / 4);
    return 0;
}

// exclusive scan of n 64-bit counts in place, total -> *total (device); `bsum` = scratch of n / SCAN_CHUNK + 1 values
static void scan_u64(hipStream_t st, unsigned long long *v, size_t n, unsigned long long *total, unsigned long long *bsum) {
    if (n <= 65536 || !bsum) { hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, v, n, total, 0ULL); return; }
    const size_t nblk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(scan_u64_blocksum_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, (const unsigned long long *)v, n, bsum);
    hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, bsum, nblk, total, 0ULL);
    hipLaunchKernelGGL(scan_u64_apply_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, v, n, (const unsigned long long *)bsum);
}

// matchSequencesSweepSparse (pbwtMatch.c:501-602).  Phase A recovers the panel's columns in original
// haplotype order on the device (the sparse cursors are BUILT from them: pbwtMatch.c:537-541 unsorts the
// panel column through a[] and gathers it into the sparse cursor's order).  Phase B runs, per batch of
// sites, the panel chain (read side, with d), the query chain, the nSparse sparse chains (build side,
// with d, each over the sites = kk mod nSparse) and one thread per query through the batch's sites.
extern "C" int pbwtamd_match_sweep_sparse(pbwtamd_engine *e, const uint8_t *pz, int64_t pnz, int N, const int32_t *pStart,
                                          int Mq, const uint8_t *qz, int64_t qnz, const int32_t *qStart, int nSparse,
                                          pbwtamd_report5_fn report, pbwtamd_match5 **recs_out, int64_t *nrecs_out,
                                          int64_t *n_nomatch, int64_t *tot_out) {
    HIPCHK(hipSetDevice(e->device));
    if ((report ? 1 : 0) + (recs_out ? 1 : 0) != 1) return fail("pbwtamd_match_sweep_sparse: exactly one of report / recs_out must be given");
    auto wall = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    const double tw0 = wall(); double tw1 = tw0, tw2 = tw0, tw3 = tw0, tw4 = tw0;
    const int nS = nSparse > 1 ? nSparse : 0;
    const int Mp = e->M, wpc = e->wpc, wpc64 = e->wpc64;
    if (nS > e->B) return fail("pbwtamd_match_sweep_sparse: nSparse %d exceeds the engine's batch of %d sites", nSparse, e->B);
    const int Bd = nS ? (e->B / nS) * nS : e->B;             // dense batch: a whole number of sparse rounds
    const int Bs = nS ? Bd / nS : 0;
    pbwtamd_engine *eq = nullptr;
    // the query cursor's chain on a stream of NORMAL priority: two high-priority streams share a hardware queue, where the panel's and
    // the queries' dependent launches take turns (measured: 320 launches per batch one after the other); on a queue of its own the
    // query chain runs beside the panel's
    hipStream_t qchain = nullptr;
    static const bool qs_own_queue = !(tune_env("PBWTAMD_QS_QCHAIN") && !atoi(tune_env("PBWTAMD_QS_QCHAIN")));
    if (qs_own_queue) HIPCHK(hipStreamCreateWithPriority(&qchain, hipStreamNonBlocking, 0));
    struct QcGuard { hipStream_t s; ~QcGuard() { if (s) (void)hipStreamDestroy(s); } } qcGuard{qchain};        // destroyed after the engine that runs on it
    CHK(pbwtamd_engine_create(&eq, e->device, Mq, e->B, (void *)qchain));
    struct EngGuard { std::vector<pbwtamd_engine *> v; ~EngGuard() { for (auto *p : v) if (p) pbwtamd_engine_destroy(p); } } guard;
    guard.v.push_back(eq);
    // the query cursor's chain: one launch per batch instead of two per round (skel_persist_kernel) — it leaves the launch stream to the panel
    static const bool qs_persist = !(tune_env("PBWTAMD_QS_PERSIST") && !atoi(tune_env("PBWTAMD_QS_PERSIST")));
    eq->persist = qs_persist;
    std::vector<pbwtamd_engine *> es((size_t)nS, nullptr);
    for (int kk = 0; kk < nS; ++kk) { CHK(pbwtamd_engine_create(&es[kk], e->device, Mp, Bs + 1, nullptr)); guard.v.push_back(es[kk]); }
    DevBufs bufs;
    Packed pk, qk;
    tw1 = wall();
    CHK(packed_upload(e, e->stream, Mp, pz, pnz, N, pk));
    CHK(packed_upload(eq, eq->stream, Mq, qz, qnz, N, qk));
    tw2 = wall();
    // ---- phase A: original-order bit columns of the whole panel ----
    uint32_t *orig = nullptr;
    if (nS) {
        CHK(bufs.alloc(&orig, (size_t)(N + 1) * wpc));
        unsigned char *dout; CHK(bufs.alloc(&dout, (size_t)e->B * Mp));
        CHK(pbwtamd_pass_begin(e, pStart, 0, N));
        for (int done = 0; done < N;) {
            const int nb = std::min(e->B, N - done), navail = std::min(nb + 1, N - done);
            CHK(packed_expand(e, e->stream, pk, Mp, done, navail, (unsigned long long *)e->cols_stage, wpc64));
            CHK(pbwtamd_pass_advance(e, e->cols_stage, wpc, nb, navail, PBWTAMD_OPT_SORTED | OPT_INTERNAL_KEEP_STATES));
            CHK(pbwtamd_sync(e));
            const int *A = ringA(e, e->ring ^ 1);
            dim3 grid(std::min(64, (Mp + BLOCK - 1) / BLOCK), nb);
            hipLaunchKernelGGL(unsort_alleles_kernel, grid, dim3(BLOCK), 0, e->stream, A, e->strideA, Mp, dout);
            dim3 g2(std::min(64, (wpc64 + WAVES - 1) / WAVES), nb);
            hipLaunchKernelGGL(bytes_to_bits_kernel, g2, dim3(BLOCK), 0, e->stream, (const unsigned char *)dout, Mp, (unsigned long long *)(orig + (size_t)done * wpc), wpc64);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(e->stream));
            done += nb;
        }
        CHK(pbwtamd_pass_end(e, PBWTAMD_OPT_SORTED));
    }
    // ---- phase B ----
    // the panel's fill (1.3 ms per 512 sites at M = 1 M) in sub-batches of 16 rounds: it runs beside the batch's own chain and the previous
    // batch's query sweep instead of between the two
    struct SubGuard { pbwtamd_engine *p; int old; ~SubGuard() { p->sub_rounds = old; } } subGuard{e, e->sub_rounds};
    static const int qs_sub = tune_env("PBWTAMD_QS_SUB") ? atoi(tune_env("PBWTAMD_QS_SUB")) : 16;
    e->sub_rounds = qs_sub;
    CHK(pbwtamd_pass_begin(e, pStart, 0, N));
    CHK(pbwtamd_pass_begin(eq, qStart, 0, N));
    std::vector<int> nTotS((size_t)nS, 0);
    for (int kk = 0; kk < nS; ++kk) { nTotS[kk] = N > kk ? (N - kk + nS - 1) / nS : 0; CHK(pbwtamd_pass_begin(es[kk], nullptr, 0, nTotS[kk])); }
    unsigned char *xq; int *invq, *rankdir, *fst[2], *dst[2], *fss[2], *dss[2]; unsigned long long *cnt, *tot; Rec5 *recs = nullptr; size_t recsCap = 0;
    const size_t BQ = (size_t)e->B * Mq;
    CHK(bufs.alloc(&xq, BQ)); CHK(bufs.alloc(&invq, BQ)); CHK(bufs.alloc(&cnt, 2 * std::max(BQ, (size_t)Mq)));
    int2 *evt; CHK(bufs.alloc(&evt, 2 * BQ));
    int *a0P, *a0Q; CHK(bufs.alloc(&a0P, (size_t)e->strideA)); CHK(bufs.alloc(&a0Q, (size_t)eq->strideA));   // first rows of a batch, kept for the emission pass
    std::vector<int *> a0S((size_t)nS, nullptr);
    for (int kk = 0; kk < nS; ++kk) CHK(bufs.alloc(&a0S[kk], (size_t)e->strideA));
    CHK(bufs.alloc(&rankdir, (size_t)e->B * (wpc64 + 1)));
    // block summaries {max d, alleles present} per 256 positions of every state of a batch: what lets the walks of reportAndUpdate skip
    // 65 536 positions per trip to memory (qs_blocksum_kernel); PBWTAMD_QS_BLOCKS=0: the walks test every position (A/B runs)
    static const bool qs_blocks = !(getenv("PBWTAMD_QS_BLOCKS") && !atoi(getenv("PBWTAMD_QS_BLOCKS")));
    const int nblk = (Mp + 255) / 256;
    int2 *bsumP[2] = {nullptr, nullptr};
    if (qs_blocks) for (int i = 0; i < 2; ++i) CHK(bufs.alloc(&bsumP[i], (size_t)e->B * nblk));
    struct BsGuard { pbwtamd_engine *p; ~BsGuard() { p->qs_bsum[0] = p->qs_bsum[1] = nullptr; p->qs_nblk = 0; } } bsGuard{e};
    // the panel's summaries are written by its own consumers (stream s2, after each sub-batch's fill) — off the sweep's critical path
    e->qs_bsum[0] = bsumP[0]; e->qs_bsum[1] = bsumP[1]; e->qs_nblk = nblk;
    std::vector<int2 *> bsumS((size_t)nS, nullptr);
    if (qs_blocks) for (int kk = 0; kk < nS; ++kk) CHK(bufs.alloc(&bsumS[kk], (size_t)(Bs + 2) * nblk));
    for (int i = 0; i < 2; ++i) {
        CHK(bufs.alloc(&fst[i], (size_t)Mq)); CHK(bufs.alloc(&dst[i], (size_t)Mq));
        CHK(bufs.alloc(&fss[i], (size_t)2 * std::max(nS, 1) * Mq)); CHK(bufs.alloc(&dss[i], (size_t)2 * std::max(nS, 1) * Mq));
    }
    CHK(bufs.alloc(&tot, (size_t)4));
    // "no match to query" events (pbwtMatch.c:405-410 / 494): the reference logs every one, in site order.  The device buffer is drained after EVERY
    // batch (sorted by site, query rank, dense before sparse = the log order), so the first NM_KEEP events kept for the caller's log are the
    // reference's first NM_KEEP lines; the count is exact beyond that.  (One batch alone would have to exceed NM_CAP events to lose that.)
    constexpr unsigned NM_CAP = 1u << 20, NM_KEEP = 1u << 16;
    int4 *nm_ev; unsigned *nm_n;
    CHK(bufs.alloc(&nm_ev, (size_t)NM_CAP)); CHK(bufs.alloc(&nm_n, (size_t)1));
    e->nomatch_events.clear();
    unsigned long long *qs_dbg = nullptr;                   // PBWTAMD_QS_DBG=<file>: per-query wave time and event count of the sweep kernel, dumped at the end
    if (tune_env("PBWTAMD_QS_DBG")) { CHK(bufs.alloc(&qs_dbg, (size_t)2 * Mq + 128)); HIPCHK(hipMemset(qs_dbg, 0, sizeof(unsigned long long) * (2 * (size_t)Mq + 128))); }
    unsigned long long *bsum; CHK(bufs.alloc(&bsum, 2 * std::max(BQ, (size_t)Mq) / SCAN_CHUNK + 2));
    std::vector<unsigned long long *> ycS((size_t)nS, nullptr); std::vector<int *> rdS((size_t)nS, nullptr);
    for (int kk = 0; kk < nS; ++kk) { CHK(bufs.alloc(&ycS[kk], (size_t)(Bs + 2) * wpc64)); CHK(bufs.alloc(&rdS[kk], (size_t)(Bs + 2) * (wpc64 + 1))); }
    QsView *dviews = nullptr; CHK(bufs.alloc(&dviews, (size_t)std::max(nS, 1)));
    std::vector<QsView> hviews_buf[2] = {std::vector<QsView>((size_t)std::max(nS, 1)), std::vector<QsView>((size_t)std::max(nS, 1))};   // per batch parity: the async upload of one batch's views may still be reading while the next batch's are filled in
    int hv_par = 0;
    // the query sweep has a stream of its own: one wave per query walks the batch's sites one dependent round trip after the
    // other (4.4 ms per 512 sites at M = 1 M, Q = 10 k: latency, the chip's bandwidth idles), so the NEXT batch's fill — queued on
    // the consumer stream s2 — runs beside it instead of behind it
    hipStream_t st = nullptr;
    {
        int prLow = 0, prHigh = 0; (void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);
        static const int qs_prio = tune_env("PBWTAMD_QS_PRIO") ? atoi(tune_env("PBWTAMD_QS_PRIO")) : 0;     // 0 low, 1 normal, 2 high
        // the sweep's ~10 000 waves live for the whole batch and slow every dependent launch of the panel's chain that has to find room beside
        // them; it has time to spare (2 ms of a 4.5 ms batch), so it may be confined to the LAST qs_cus CUs of the device (0 = no mask)
        static const int qs_cus = tune_env("PBWTAMD_QS_CUS") ? atoi(tune_env("PBWTAMD_QS_CUS")) : QS_SWEEP_CUS;
        int ncu_dev = 0; (void)hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, e->device);
        if (qs_cus > 0 && qs_cus < ncu_dev && ncu_dev <= 256) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = ncu_dev - qs_cus; i < ncu_dev; ++i) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
        }
        if (!st) HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, qs_prio == 2 ? prHigh : qs_prio == 1 ? 0 : prLow));
    }
    struct StGuard { hipStream_t s; ~StGuard() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } } stGuard{st};
    HIPCHK(hipMemsetAsync(fst[0], 0, sizeof(int) * (size_t)Mq, st));       // calloc'ed f[], d[], ff[][], dd[][] (pbwtMatch.c:512-523)
    HIPCHK(hipMemsetAsync(dst[0], 0, sizeof(int) * (size_t)Mq, st));
    HIPCHK(hipMemsetAsync(fss[0], 0, sizeof(int) * (size_t)2 * std::max(nS, 1) * Mq, st));
    HIPCHK(hipMemsetAsync(dss[0], 0, sizeof(int) * (size_t)2 * std::max(nS, 1) * Mq, st));
    HIPCHK(hipMemsetAsync(tot, 0, 4 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(nm_n, 0, sizeof(unsigned), st));
    // the records on the host: ONE malloc'ed buffer that grows geometrically and is handed to the caller as it is (recs_out; pbwtamd_free) —
    // at 10^7 records a std::vector cost a zero-fill per growth and a copy out at the end
    struct HostRecs { pbwtamd_match5 *p = nullptr; size_t n = 0, cap = 0; ~HostRecs() { free(p); } } all;
    auto ensure_recs = [&](size_t total) -> int {
        if (total <= recsCap) return 0;
        recsCap = total + total / 4 + 1024;
        return bufs.alloc(&recs, recsCap);
    };
    auto deliver = [&](size_t total) -> int {
        if (!total) return 0;
        const size_t old = all.n;
        if (old + total > all.cap) {
            const size_t cap = std::max(old + total, all.cap + all.cap / 2 + 4096);
            pbwtamd_match5 *q = (pbwtamd_match5 *)realloc(all.p, cap * sizeof(pbwtamd_match5));
            if (!q) return fail("pbwtamd_match_sweep_sparse: out of host memory");
            all.p = q; all.cap = cap;
        }
        HIPCHK(hipMemcpyAsync(all.p + old, recs, total * sizeof(Rec5), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        all.n = old + total;
        if (report) { for (size_t r = old; r < old + total; ++r) report(all.p[r].ai, all.p[r].bi, all.p[r].start, all.p[r].end, all.p[r].sparse); all.n = old; }
        return 0;
    };
    unsigned long long *h_total = nullptr; hipEvent_t evTotal = nullptr;      // the batch's record count comes back through pinned memory + an event
    HIPCHK(hipHostMalloc((void **)&h_total, 2 * sizeof(unsigned long long), hipHostMallocDefault));     // [1]: the batch's no-match event count
    HIPCHK(hipEventCreateWithFlags(&evTotal, hipEventDisableTiming));
    hipEvent_t evCols = nullptr; HIPCHK(hipEventCreateWithFlags(&evCols, hipEventDisableTiming));
    struct TotGuard { unsigned long long *p; hipEvent_t ev, ev2; ~TotGuard() { if (ev) (void)hipEventDestroy(ev); if (ev2) (void)hipEventDestroy(ev2); if (p) (void)hipHostFree(p); } } totGuard{h_total, evTotal, evCols};
    int cur = 0;
    static const bool trace_qs = getenv("PBWTAMD_TRACE_QS") != nullptr;
    double tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    double tmark = now();
    auto lap = [&](int i) { if (trace_qs) { const double t = now(); tph[i] += t - tmark; tmark = t; } };
    const int qblocks = (Mq + BLOCK - 1) / BLOCK;          // thread per query (qs_unsort)
    const int qwaves = (Mq + WAVES - 1) / WAVES;            // wave per query (sweep, tails)
    // the chains of the batch starting at site `at` (panel, queries, sparse cursors): enqueued only — batch b+1's chains run
    // while batch b's query sweep does (the sweep's stream records the consumer events the chains wait for before they
    // overwrite the ring the sweep reads)
    // The panel's decoded columns alternate between the two halves of the staging buffer, so that batch b+1's columns, rank
    // directories and skeleton keys (skel_keys_sorted_kernel: 0.9 ms per batch at M = 1 M, a function of the columns alone) are
    // derived on a stream of their own WHILE batch b's chain runs, instead of in front of batch b+1's chain.
    hipStream_t pre = nullptr; hipEvent_t evPre[2] = {nullptr, nullptr};
    {
        int prLow = 0, prHigh = 0; (void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);
        HIPCHK(hipStreamCreateWithPriority(&pre, hipStreamNonBlocking, prLow));
        for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&evPre[i], hipEventDisableTiming));
    }
    struct PreGuard { hipStream_t s; hipEvent_t *ev; pbwtamd_engine *p; ~PreGuard() { p->evPreKeys = nullptr; if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } for (int i = 0; i < 2; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]); } } preGuard{pre, evPre, e};
    int *rdPre = nullptr; CHK(bufs.alloc(&rdPre, (size_t)(e->B + 2) * (wpc64 + 1)));
    static const bool qs_prefetch = !(tune_env("PBWTAMD_QS_PREFETCH") && !atoi(tune_env("PBWTAMD_QS_PREFETCH")));
    // three staging slots in rotation: batch b's columns are read by the sweep and the emission pass of batch b (this loop's iteration b, which
    // ends with the host waiting for both), by the chain + fill of batch b (enqueued in iteration b-1), and written by the prefetch enqueued in
    // iteration b-2 — the slot's previous tenant, batch b-3, was finished with in iteration b-3.  (Two halves needed a 64 MB copy per batch at
    // M = 1 M to keep the sweep's columns alive: 0.1-0.6 ms of a 6.6 ms batch beside the other kernels.)
    uint32_t *stage_extra = nullptr; CHK(bufs.alloc(&stage_extra, ((size_t)e->B + 8) * wpc));
    auto stage_half = [&](int at) -> uint32_t * { const int sl = (at / Bd) % 3; return sl == 2 ? stage_extra : e->cols_stage + (size_t)sl * ((size_t)e->B + 8) * wpc; };
    int pre_at = -1;                                       // the batch whose columns + keys are prepared (or being prepared) on `pre`
    // prepare the batch starting at `at`, which will run in ring `ring`: decode + rank directories + keys of every round
    auto prefetch = [&](int at, int ring) -> int {
        const int nb = std::min(Bd, N - at);
        if (!qs_prefetch || nb % 8 || !e->skel) return 0;  // only the skeleton path consumes prepared keys
        const int navail = std::min(nb + 1, N - at);
        unsigned long long *yc = (unsigned long long *)stage_half(at);
        HIPCHK(hipStreamWaitEvent(pre, evCols, 0));         // the sweep's copy out of this half (two batches ago) has been enqueued before this call
        CHK(packed_expand(e, pre, pk, Mp, at, navail, yc, wpc64));
        hipLaunchKernelGGL(qs_rankdir_kernel, dim3(nb), dim3(BLOCK), 0, pre, (const unsigned long long *)yc, wpc64, Mp, rdPre);
        hipLaunchKernelGGL(skel_keys_sorted_kernel, dim3((Mp + BLOCK - 1) / BLOCK, nb / 8), dim3(BLOCK), 0, pre, (const unsigned long long *)yc, wpc64,
                           (const int *)rdPre, Mp, e->keysR[ring], (size_t)e->Mpad);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(evPre[(at / Bd) & 1], pre));
        pre_at = at;
        return 0;
    };
    // the panel's fill writes d only; the ids of reported positions come from the next skeleton state (qss_emit_kernel)
    static const bool qs_lazy_ids = !(tune_env("PBWTAMD_QS_LAZY_IDS") && !atoi(tune_env("PBWTAMD_QS_LAZY_IDS")));
    auto enqueue_chains = [&](int at) -> int {
        const int nb = std::min(Bd, N - at);
        const int navail = std::min(nb + 1, N - at);
        uint32_t *stage = stage_half(at);
        if (pre_at == at) { HIPCHK(hipStreamWaitEvent(e->stream, evPre[(at / Bd) & 1], 0)); e->evPreKeys = evPre[(at / Bd) & 1]; }   // skel_prepare skips the keys
        else CHK(packed_expand(e, e->stream, pk, Mp, at, navail, (unsigned long long *)stage, wpc64));
        // the query cursor first: its chain is one launch (skel_persist_kernel) — enqueued behind the panel's 192 throttled launches it would
        // only start when the host gets there, near the END of the panel's chain (measured: 3.3 ms into a 5 ms batch)
        CHK(packed_expand(eq, eq->stream, qk, Mq, at, navail, (unsigned long long *)eq->cols_stage, eq->wpc64));
        CHK(pbwtamd_pass_advance(eq, eq->cols_stage, eq->wpc, nb, navail, PBWTAMD_OPT_SORTED | OPT_INTERNAL_KEEP_STATES));
        CHK(pbwtamd_pass_advance(e, stage, wpc, nb, navail, PBWTAMD_OPT_SORTED | PBWTAMD_OPT_WITH_D | OPT_INTERNAL_KEEP_STATES | (qs_lazy_ids ? OPT_INTERNAL_D_ONLY : 0u)));
        e->evPreKeys = nullptr;
        for (int kk = 0; kk < nS; ++kk) {                  // the sparse cursors' steps that fall into this batch
            const int ns = nb > kk ? (nb - kk + nS - 1) / nS : 0;
            if (!ns) continue;
            pbwtamd_engine *s = es[kk];
            const int left = nTotS[kk] - at / nS;          // sparse sites from this batch's first one on
            const int nav = std::min(std::min(ns + 8, left), s->B + 8);
            HIPCHK(hipMemcpy2DAsync(s->cols_stage, (size_t)wpc * 4, orig + (size_t)(at + kk) * wpc, (size_t)nS * wpc * 4, (size_t)wpc * 4, (size_t)nav,
                                    hipMemcpyDeviceToDevice, s->stream));
            CHK(pbwtamd_pass_advance(s, s->cols_stage, wpc, ns, nav, PBWTAMD_OPT_WITH_D | OPT_INTERNAL_KEEP_STATES));
        }
        return 0;
    };
    tw3 = wall();
    if (N > 0) CHK(enqueue_chains(0));
    if (N > Bd) CHK(prefetch(Bd, 1));
    for (int done = 0; done < N;) {
        const int nb = std::min(Bd, N - done);
        std::vector<QsView> &hviews = hviews_buf[hv_par]; hv_par ^= 1;
        for (int kk = 0; kk < nS; ++kk) hviews[kk] = QsView{nullptr, nullptr, 0, 0, nullptr, nullptr, done / nS, nullptr, nullptr, 0};
        lap(0);
        CHK(pbwtamd_sync(e));
        CHK(pbwtamd_sync(eq));
        lap(1);
        const int *A = ringA(e, e->ring ^ 1), *D = ringD(e, e->ring ^ 1), *AQ = ringA(eq, eq->ring ^ 1);
        const bool lazy = qs_lazy_ids && e->ring_skel[e->ring ^ 1];   // this batch's slots 1..7 mod 8 hold no ids
        const int *Anext = ringA(e, e->ring);                // the state after the batch's last site (written by the batch's last launch; the next writer is the chain two batches on, which waits for this stream)
        // the panel's sorted bit columns of this batch are the decoded input columns themselves (read side), read where the decode left them
        const unsigned long long *ycB = (const unsigned long long *)stage_half(done);
        HIPCHK(hipEventRecord(evCols, st));
        hipLaunchKernelGGL(qs_rankdir_kernel, dim3(nb), dim3(BLOCK), 0, st, ycB, wpc64, Mp, rankdir);
        if (bsumP[0]) {                                     // states the batch's consumers did not summarise (batches outside the skeleton path run no consumers for this option set)
            const int rr = e->ring ^ 1, have = std::min(e->qs_bsum_sites[rr], nb);
            if (have < nb) hipLaunchKernelGGL(qs_blocksum_kernel, dim3((nblk + 4 * WAVES - 1) / (4 * WAVES), nb - have), dim3(BLOCK), 0, st, D + (size_t)have * e->strideD, e->strideD,
                                              ycB + (size_t)have * wpc64, wpc64, Mp, nblk, bsumP[rr] + (size_t)have * nblk);
        }
        hipLaunchKernelGGL(qs_unsort_kernel, dim3(std::min(qblocks, 64), nb), dim3(BLOCK), 0, st, AQ, eq->strideA, Mq, xq, invq);
        for (int kk = 0; kk < nS; ++kk) {
            const int ns = nb > kk ? (nb - kk + nS - 1) / nS : 0;
            if (!ns) continue;
            pbwtamd_engine *s = es[kk];
            CHK(pbwtamd_sync(s));                          // incl. the fill of the skeleton path
            const int *As = ringA(s, s->ring ^ 1), *Ds = ringD(s, s->ring ^ 1);
            dim3 gs(std::min(64, (wpc64 + WAVES - 1) / WAVES), ns);
            hipLaunchKernelGGL(tags_to_bits_kernel, gs, dim3(BLOCK), 0, st, As, s->strideA, Mp, ycS[kk], wpc64);
            hipLaunchKernelGGL(qs_rankdir_kernel, dim3(ns), dim3(BLOCK), 0, st, (const unsigned long long *)ycS[kk], wpc64, Mp, rdS[kk]);
            if (bsumS[kk]) hipLaunchKernelGGL(qs_blocksum_kernel, dim3((nblk + 4 * WAVES - 1) / (4 * WAVES), ns), dim3(BLOCK), 0, st, Ds, s->strideD, (const unsigned long long *)ycS[kk], wpc64, Mp, nblk, bsumS[kk]);
            HIPCHK(hipMemcpyAsync(a0S[kk], As, sizeof(int) * (size_t)Mp, hipMemcpyDeviceToDevice, st));
            hviews[kk] = QsView{As, Ds, s->strideA, s->strideD, ycS[kk], rdS[kk], done / nS, a0S[kk], bsumS[kk], nblk};
        }
        if (nS) HIPCHK(h2d_async(dviews, hviews.data(), sizeof(QsView) * (size_t)nS, st));
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * 2 * (size_t)nb * Mq, st));
        QssArgs g;
        HIPCHK(hipMemcpyAsync(a0P, A, sizeof(int) * (size_t)Mp, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(a0Q, AQ, sizeof(int) * (size_t)Mq, hipMemcpyDeviceToDevice, st));
        g.dense = QsView{A, D, e->strideA, e->strideD, ycB, rankdir, 0, a0P, bsumP[e->ring ^ 1], nblk}; g.sparse = dviews; g.wpc64 = wpc64; g.nS = nS;
        g.xq = xq; g.invq = invq; g.Mp = Mp; g.Mq = Mq; g.kbase = done; g.nsites = nb;
        g.f_in = fst[cur]; g.dq_in = dst[cur]; g.f_out = fst[cur ^ 1]; g.dq_out = dst[cur ^ 1];
        g.fs_in = fss[cur]; g.ds_in = dss[cur]; g.fs_out = fss[cur ^ 1]; g.ds_out = dss[cur ^ 1];
        g.cnt = cnt; g.recs = nullptr; g.tot = tot;
        g.nm_ev = nm_ev; g.nm_n = nm_n; g.nm_cap = NM_CAP; g.evt = evt;
        g.q_lo = e->q_lo; g.q_hi = e->q_hi; g.dbg = qs_dbg;
        // a wave lives for the whole batch here (one query, site after site): at full occupancy the next batch's chain kernels,
        // enqueued below to run beside it, would find no wave slot until it ends.  26 KB of (unused) dynamic LDS per workgroup
        // holds the sweep to 6 of the 8 wave slots per SIMD.
        static const int qs_lds_kb = tune_env("PBWTAMD_QS_LDS") ? atoi(tune_env("PBWTAMD_QS_LDS")) : 26;
        static const int qs_qpw = tune_env("PBWTAMD_QS_QPW") ? std::max(1, atoi(tune_env("PBWTAMD_QS_QPW"))) : QS_QUERIES_PER_WAVE;
        hipLaunchKernelGGL((qss_sweep_kernel<0>), dim3((qwaves + qs_qpw - 1) / qs_qpw), dim3(BLOCK), (size_t)qs_lds_kb * 1024, st, g);
        scan_u64(st, cnt, 2 * (size_t)nb * Mq, tot + 3, bsum);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_total, tot + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));   // read back before the next batch's fill queues on this stream
        HIPCHK(hipMemcpyAsync(h_total + 1, nm_n, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(evTotal, st));
        {   // this batch's rings are read by the kernels just enqueued on `st`: the next batch's chains wait for them
            auto mark = [&](pbwtamd_engine *x) -> int { const int r = x->ring ^ 1; HIPCHK(hipEventRecord(x->evCons[r], st)); x->consRecorded[r] = true; return 0; };
            CHK(mark(e)); CHK(mark(eq));
            for (int kk = 0; kk < nS; ++kk) if (hviews[kk].A) CHK(mark(es[kk]));
        }
        static const bool qs_serial = tune_env("PBWTAMD_QS_SERIAL") != nullptr;                             // measurement: the sweep alone on the device, the next batch's chains after it
        if (qs_serial) HIPCHK(hipEventSynchronize(evTotal));
        if (done + nb < N) {                                 // runs beside the sweep; ring pointers of THIS batch were taken above
            HIPCHK(hipStreamWaitEvent(e->stream, evCols, 0));
            CHK(enqueue_chains(done + nb));
            {   // ... and the batch after it is prepared meanwhile
                const int at2 = done + nb + std::min(Bd, N - (done + nb));
                if (at2 < N) CHK(prefetch(at2, (at2 / Bd) & 1));
            }
            CHK(flush_pending(e)); CHK(flush_pending(eq));      // their fills queue behind the sweep on the consumer stream instead of waiting for the next sync
            for (int kk = 0; kk < nS; ++kk) CHK(flush_pending(es[kk]));
        }
        lap(2);
        HIPCHK(hipEventSynchronize(evTotal));
        const unsigned long long total = *h_total;
        if (const unsigned nev_b = std::min(*reinterpret_cast<unsigned *>(h_total + 1), NM_CAP)) {       // this batch's no-match events, in log order, while fewer than NM_KEEP are kept
            if (e->nomatch_events.size() / 4 < NM_KEEP) {
                std::vector<int4> ev(nev_b);
                HIPCHK(hipMemcpyAsync(ev.data(), nm_ev, sizeof(int4) * nev_b, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                std::sort(ev.begin(), ev.end(), [](const int4 &a, const int4 &b) { return a.x != b.x ? a.x < b.x : a.y != b.y ? a.y < b.y : (a.w >> 1) < (b.w >> 1); });
                for (const int4 &v : ev) {
                    if (e->nomatch_events.size() / 4 >= NM_KEEP) break;
                    e->nomatch_events.push_back(v.z); e->nomatch_events.push_back(v.w & 1); e->nomatch_events.push_back(v.x); e->nomatch_events.push_back((v.w >> 1) | (e->q_part ? (v.y << 1) : 0));
                }
            }
            HIPCHK(hipMemsetAsync(nm_n, 0, sizeof(unsigned), st));
        }
        lap(3);
        if (total) {
            CHK(ensure_recs((size_t)total));
            QssEmitArgs em;                                 // expand the event descriptors of the counting pass (the walks are not repeated)
            em.off = cnt; em.total = tot + 3; em.evt = evt; em.nslots = 2 * (size_t)nb * Mq;
            em.dense = g.dense; em.sparse = dviews; em.nS = std::max(nS, 1);
            em.AQ = AQ; em.strideAQ = eq->strideA; em.AQ0 = a0Q; em.Mq = Mq; em.kbase = done; em.recs = recs; em.emit_rank = e->q_part ? 1 : 0;
            em.lazy = lazy ? 1 : 0; em.nsites = nb; em.wpc64 = wpc64; em.Anext = Anext;
            const size_t ewaves = (em.nslots + 63) / 64;
            hipLaunchKernelGGL(qss_emit_kernel, dim3((unsigned)((ewaves + WAVES - 1) / WAVES)), dim3(BLOCK), 0, st, em);
            HIPCHK(hipGetLastError());
            CHK(deliver((size_t)total));
        }
        lap(4);
        cur ^= 1;
        done += nb;
    }
    tw4 = wall();
    if (trace_qs) fprintf(stderr, "pbwt_amd query sweep outside the loop (s): engines %.4f  upload of the packed panels %.4f  buffers + phase A %.4f  loop %.4f\n", tw1 - tw0, tw2 - tw1, tw3 - tw2, tw4 - tw3);
    if (trace_qs) fprintf(stderr, "pbwt_amd query sweep host phases (s): enqueue chains %.4f  wait chains+fill %.4f  enqueue sweep %.4f  wait count %.4f  emit+deliver %.4f\n", tph[0], tph[1], tph[2], tph[3], tph[4]);
    // ---- matches still open at N: the panel cursor for every query, then each sparse cursor in turn (pbwtMatch.c:577-594) ----
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(eq->stream));
    for (int c = -1; c < nS; ++c) {
        pbwtamd_engine *s = c < 0 ? e : es[c];
        if (c >= 0) CHK(pbwtamd_sync(s));
        const int *A = ringA(s, s->ring), *D = ringD(s, s->ring), *AQ = ringA(eq, eq->ring);
        const int *fp = c < 0 ? fst[cur] : fss[cur] + (size_t)c * Mq, *dp = c < 0 ? dst[cur] : dss[cur] + (size_t)c * Mq;
        hipLaunchKernelGGL((qss_tail_kernel<0>), dim3(qwaves), dim3(BLOCK), 0, st, A, D, AQ, Mp, Mq, N, nS, std::max(c, 0), c >= 0 ? 1 : 0, fp, dp, cnt, (Rec5 *)nullptr, tot, e->q_lo, e->q_hi, e->q_part ? 1 : 0);
        hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, cnt, (size_t)Mq, tot + 3, 0ULL);
        HIPCHK(hipGetLastError());
        unsigned long long total = 0;
        HIPCHK(hipMemcpyAsync(&total, tot + 3, sizeof total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        CHK(ensure_recs((size_t)total));
        hipLaunchKernelGGL((qss_tail_kernel<1>), dim3(qwaves), dim3(BLOCK), 0, st, A, D, AQ, Mp, Mq, N, nS, std::max(c, 0), c >= 0 ? 1 : 0, fp, dp, cnt, recs, tot, e->q_lo, e->q_hi, e->q_part ? 1 : 0);
        HIPCHK(hipGetLastError());
        CHK(deliver((size_t)total));
    }
    if (qs_dbg) {
        std::vector<unsigned long long> h((size_t)2 * Mq + 128);
        HIPCHK(hipMemcpy(h.data(), qs_dbg, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(tune_env("PBWTAMD_QS_DBG"), "w")) {
            for (int q = 0; q < Mq; ++q) fprintf(f, "%d %llu %llu\n", q, h[2 * (size_t)q], h[2 * (size_t)q + 1]);
            for (int b = 0; b < 64; ++b) fprintf(f, "%d %llu %llu\n", -1 - b, h[2 * (size_t)Mq + b], h[2 * (size_t)Mq + 64 + b]);     // per batch: slowest wave (ticks), most events of one query
            fclose(f);
        }
    }
    unsigned long long htot[4];
    HIPCHK(hipMemcpy(htot, tot, sizeof htot, hipMemcpyDeviceToHost));
    if (tot_out) { tot_out[0] = (int64_t)htot[0]; tot_out[1] = (int64_t)htot[1]; }
    if (n_nomatch) *n_nomatch = (int64_t)htot[2];
    CHK(pbwtamd_pass_end(e, 0));
    CHK(pbwtamd_pass_end(eq, 0));
    for (int kk = 0; kk < nS; ++kk) CHK(pbwtamd_pass_end(es[kk], 0));
    if (recs_out) {
        if (!all.p && !(all.p = (pbwtamd_match5 *)malloc(sizeof(pbwtamd_match5)))) return fail("pbwtamd_match_sweep_sparse: out of host memory");
        *recs_out = all.p; *nrecs_out = (int64_t)all.n;
        all.p = nullptr;                                    // the caller's now
    }
    if (trace_qs) fprintf(stderr, "pbwt_amd query sweep after the loop (s): tails + pass_end + copy out %.4f; whole call %.4f\n", wall() - tw4, wall() - tw0);
    return 0;
}
