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
#include <chrono>
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
struct Pending {
    bool valid = false; int ring = 0, kbase = 0, nb = 0; unsigned opts = 0; bool skel = false; bool sharded = false;
    bool shard_tables = false;              // sharded batch whose tile tables were pulled from the chain (rows per tile, no pair rows)
    bool early = false;
    int flushed = 0;                        // leading sites whose consumers are enqueued already
    const uint32_t *cols = nullptr;         // the batch's bit columns
};

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
    int2 *aggxR[2] = {nullptr, nullptr};                     // per ring and round: the scan workgroups' exclusive rows (kept for the consumers)
    int2 *peerS[2][SHARD_MAX] = {}, *peerX[2][SHARD_MAX] = {};   // every rank's saveR blocks and aggxR rows as mapped here
    int *peerA[SHARD_MAX] = {}, *peerD[SHARD_MAX] = {}; unsigned char *peerK[2][SHARD_MAX] = {};   // skeleton rings and key rows of every rank (own = this rank's)
    bool connected = false;
    int *SA = nullptr, *SD = nullptr; size_t nslot = 0;       // the skeleton ring: 2 rings of B/8+1 slots, a and d of the states 0, 8, 16, ... of a batch
    int2 *tbl = nullptr, *scan = nullptr; int *total = nullptr;              // chain scratch, rows indexed by global tile
    unsigned long long *agg = nullptr; unsigned *cnt = nullptr; unsigned cntEpoch = 0; int2 *aggx = nullptr;   // two-level tile scan of the rank's tiles; aggx: one exclusive row per scan workgroup (skel_k2s_local_kernel)
    unsigned e1 = 0, e2 = 0, e3 = 0;                          // epochs of f1 / f2 / f3
    int2 *ctbl = nullptr; unsigned long long *cagg = nullptr; unsigned *ccnt = nullptr; unsigned cEpoch = 0;   // consumer stream: hist rows + two-level scan state
    std::vector<long long> blkSite0, blkSites; unsigned long long *blkEnd = nullptr; size_t blkCap = 0;          // pack3 blocks this rank wrote
    bool full_state = true;                                   // slot 0 of the current ring is complete on this rank (pass start, after a replicated batch)
};

// the records of a sweep on the host: ONE malloc'ed buffer that grows geometrically (realloc: no value-initialisation pass over gigabytes, as std::vector::resize
// makes) and is handed to the caller as it is (pbwtamd_free) — 10^8 records of configs[1] are 1.6 GB, every extra pass over them a third of a second
struct RecBuf {
    pbwtamd_match *p = nullptr; size_t n = 0, cap = 0;
    ~RecBuf() { free(p); }
    pbwtamd_match *grow(size_t add) {                       // room for `add` more records behind the n there are; nullptr: out of memory
        if (n + add > cap) {
            const size_t want = std::max(n + add, cap + cap / 2 + 1024);
            pbwtamd_match *q = (pbwtamd_match *)realloc(p, want * sizeof(pbwtamd_match));
            if (!q) return nullptr;
            p = q; cap = want;
        }
        return p + n;
    }
    pbwtamd_match *release() { pbwtamd_match *q = p ? p : (pbwtamd_match *)malloc(sizeof(pbwtamd_match)); p = nullptr; n = cap = 0; return q; }
};
struct pbwtamd_engine {
    int device = 0, M = 0, Mpad = 0, wpc = 0, wpc64 = 0, W = 0, wpad = 0, E = 4, T = 1024, B = 0;
    bool pinned = false;                    // this engine holds a count of pin_first_engine
    hipStream_t stream = nullptr; bool own_stream = false;   // the launch chain
    hipStream_t s2 = nullptr;                                 // batch consumers
    hipEvent_t evChain[2] = {nullptr, nullptr}, evCons[2] = {nullptr, nullptr}; bool consRecorded[2] = {false, false}, chainRecorded[2] = {false, false};
    int2 *qs_bsum[2] = {nullptr, nullptr}; int qs_nblk = 0;   // query sweep: per ring, block summaries of every state of the batch (qs_blocksum_kernel), written by the batch's consumers
    int qs_bsum_sites[2] = {0, 0};          // ... and how many leading sites of the ring's batch they have summarised so far
    SkArgs *margs = nullptr, *margs_host = nullptr; size_t margs_cap = 0; int margs_half = 0; hipEvent_t evMargs[2] = {nullptr, nullptr};   // pbwtamd_pass_advance_many (panel 0 owns them)
    // the one-launch round (skel_onepass_kernel; PBWTAMD_ONEPASS): tagged row / group-row granules, tiles per group, launches so far (the tag)
    bool op_ordered = false;               // one-launch round with tile = workgroup index (no XCD-contiguous dealing): see skel_round_args
    hipStream_t s2_single = nullptr; bool s2_many = false;      // pass_advance_many: the consumer stream confined to more CUs (many_consumer_stream); the first one kept for destroy
    bool op_merged = false; int op_fk = 1;  // the row out of the rank's chunk tables (256-position tiles; PBWTAMD_ONEPASS_MERGED); folder copies per group (PBWTAMD_ONEPASS_FOLDERS_K)
    bool op_both = false;                   // ... and a tile polls both look-back levels in one round trip (512-position tiles; PBWTAMD_ONEPASS_BOTH)
    bool op_folders = false;                // one-launch round: a folder workgroup per group publishes the group's aggregate (PBWTAMD_ONEPASS_FOLDERS=0: the group's last tile does)
    // (round 6) the scanner form of the one-launch round (wide panels): op_nscan scanner workgroups in front of the tiles, op_scanl[tile][key] their local prefixes
    bool op_scan = false; int op_nscan = 0; unsigned long long *op_scanl = nullptr;
    // where this stream's workgroups land (xcd_probe_kernel, read once at creation): bit x = XCD x takes workgroups; xcd_rr: workgroup b runs on XCD b mod 8 of eight
    unsigned xcd_mask = 0; bool xcd_rr = false;
    long long op_cap = 0;                   // workgroups of the one-launch kernel the device holds at once (occupancy x CUs; 0: not a one-launch engine)
    // (round 6) the NEXT batch's preparation — transposed panel, key totals — on the consumers' stream beside this batch's chain (skel_prepare_ahead): what it was found for
    hipEvent_t evPrep[2] = {nullptr, nullptr}; bool prep_valid[2] = {false, false}; const uint32_t *prep_cols[2] = {nullptr, nullptr};
    int prep_nb[2] = {0, 0}, prep_k[2] = {0, 0}, prep_navail[2] = {0, 0};
    hipStream_t h2d_stream = nullptr; hipEvent_t evCopy[2] = {nullptr, nullptr};     // ... and the stream their copies to the device ride, one batch ahead of the chain
    hipEvent_t evD2H[2] = {nullptr, nullptr};   // d2h_staged: a piece has landed in its pinned buffer
    void *h_stage[2] = {nullptr, nullptr};  // pinned host staging of pbwtamd_build (two batches of columns), allocated by the first call that copies from pageable memory
    unsigned *team_host = nullptr;          // pinned: the tickets and the error word behind the last team launch
    bool team_broken = false;               // a team of the team-persistent chain did not fill once: three launches per round from then on
    bool onepass = false; unsigned long long *op_rows = nullptr, *op_grows = nullptr; int op_g1 = 0; unsigned op_epoch = 0; unsigned long long *op_prof = nullptr;
    unsigned long long *teamprof = nullptr;                 // PBWTAMD_TEAM_PROF=1: member 0's wall-clock stamps per round and phase
    unsigned *teamctl = nullptr; unsigned team_round = 0; int team_cap = 0;   // team-persistent chain (skel_team_kernel): tickets + flag words per XCD, barriers passed so far (the first engine of a group owns them)
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
    bool k2local = false; int k2tpw = 32; size_t aggx_off = 0;   // wide panels: local-prefix scan (skel_k2_local_kernel), rows per scan workgroup, offset of the aggregate rows in a round's block of saveR (int2)
    int Wt = 0, skEPT = 4;                  // skeleton tiles: 256*skEPT positions, Wt of them; PBWTAMD_SKT=512|1024
    int skn_maxw = 48;                      // two-launch round (rank scans the tile table itself) up to this many tiles (measured: 5 k -28 %, 8 k -26 %, 10 k -9 %, 12 k -6 %, 16 k 0); PBWTAMD_SKN_MAXW
    bool summ_pair = false;                 // format of the current tile summaries (two-site keys or single site)
    uint32_t *zerocol = nullptr; long long sites_done = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t ev_used = 0; long long launches = 0;
    // record sink for pass_advance (host-buffer entry points)
    RecBuf *rec_sink = nullptr; pbwtamd_report_fn rec_cb = nullptr;
    int longL = 0;                          // L of the -longWithin consumer (PBWTAMD_OPT_LONG_RECS)
    int *ystale = nullptr;                  // copy of the previous state's tagged a, for the k == N quirk of -longWithin
    std::vector<int32_t> nomatch_events;    // (jj, x, k[, isSparse]) of the last query sweep, in the reference's log order
    ShardCtx *sh = nullptr;          // position sharding across GPUs (pbwt_shard.inc): this engine is one rank of a panel
};

// ------------------------------------------------------------------------------------ CPU placement of the launching thread
// About one fresh process in five ran the 100 k chain 10-11 % slower as a whole (every launch 4.3 instead of 3.9 us): it follows where the launching thread and
// the HIP runtime's helper threads run (tools/slowmode.sh: 5 of 16 unpinned processes slow, 0 of 20 under taskset on the physical cores of one socket), and an
// affinity set AFTER the runtime's threads exist does not confine them.  So the first pbwtamd_engine_create of a process — before its first HIP call — confines
// the calling thread, and with it every thread the runtime creates, to the physical cores (one hardware thread per core) of the NUMA node the GPU hangs off; the
// calling thread's own mask is restored when the process's last engine is destroyed.  PBWTAMD_PIN=0: off.  (A host that initialised HIP before — PyTorch — pins
// itself first: pbwt_amd/pin.py, as bench.py does.)
#include <sched.h>
#include <dirent.h>
#include <unistd.h>
#include <sys/syscall.h>
static std::mutex g_pin_mu;
static int g_pin_engines = 0; static bool g_pin_active = false; static cpu_set_t g_pin_old; static pid_t g_pin_tid = 0;      // (the mask goes back to the THREAD that was narrowed, whoever destroys the last engine)
// the physical card behind HIP device `device`: HIP_VISIBLE_DEVICES, then ROCR_VISIBLE_DEVICES, when they are plain index lists (anything else: -1, no pinning by card)
static int visible_to_physical(int device) {
    int idx = device;
    for (const char *name : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"}) {
        const char *v = getenv(name);
        if (!v || !*v) continue;
        std::vector<int> list;
        for (const char *p = v; *p;) { char *end; const long x = strtol(p, &end, 10); if (end == p || (*end && *end != ',')) return -1; list.push_back((int)x); p = *end ? end + 1 : end; }
        if (idx < 0 || idx >= (int)list.size()) return -1;
        idx = list[idx];
    }
    return idx;
}
static bool read_small(const char *path, char *buf, size_t n) { FILE *f = fopen(path, "r"); if (!f) return false; const bool ok = fgets(buf, (int)n, f) != nullptr; fclose(f); return ok; }
static void cpulist_to_set(const char *txt, cpu_set_t *set) {
    CPU_ZERO(set);
    for (const char *p = txt; p && *p;) {
        char *end; const long a = strtol(p, &end, 10); if (end == p) break;
        long b = a; if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, set);
        p = (*end == ',') ? end + 1 : end; if (*end != ',') break;
    }
}
static void pin_first_engine(int device) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_engines++ > 0) return;
    if (const char *s = getenv("PBWTAMD_PIN")) if (!atoi(s)) return;
    int node = -1, seen = 0;
    device = visible_to_physical(device);
    if (device < 0) return;
    if (DIR *d = opendir("/sys/class/drm")) {               // the device-th AMD card (PCI vendor 0x1002), in name order as far as readdir gives it
        std::vector<std::string> cards;
        while (dirent *de = readdir(d)) if (!strncmp(de->d_name, "card", 4) && !strchr(de->d_name, '-')) cards.push_back(de->d_name);
        closedir(d);
        std::sort(cards.begin(), cards.end());
        for (const std::string &c : cards) {
            char buf[64];
            if (!read_small(("/sys/class/drm/" + c + "/device/vendor").c_str(), buf, sizeof buf) || strncmp(buf, "0x1002", 6)) continue;
            if (seen++ == device || node < 0) { if (read_small(("/sys/class/drm/" + c + "/device/numa_node").c_str(), buf, sizeof buf)) node = atoi(buf); }
            if (seen > device) break;
        }
    }
    cpu_set_t cand; char txt[4096];
    bool have = false;
    if (node >= 0) { char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node); if (read_small(path, txt, sizeof txt)) { cpulist_to_set(txt, &cand); have = CPU_COUNT(&cand) > 0; } }
    if (!have && read_small("/sys/devices/system/cpu/online", txt, sizeof txt)) { cpulist_to_set(txt, &cand); have = CPU_COUNT(&cand) > 0; }
    if (!have || sched_getaffinity(0, sizeof g_pin_old, &g_pin_old) != 0) return;
    cpu_set_t want; CPU_ZERO(&want);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &cand) || !CPU_ISSET(c, &g_pin_old)) continue;
        char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        cpu_set_t sib; bool first = true;
        if (read_small(path, txt, sizeof txt)) { cpulist_to_set(txt, &sib); for (int q = 0; q < c; ++q) if (CPU_ISSET(q, &sib)) { first = false; break; } }
        if (first) CPU_SET(c, &want);                       // one hardware thread per core
    }
    if (CPU_COUNT(&want) < 2) return;                       // (a narrow cpuset: leave it alone)
    if (sched_setaffinity(0, sizeof want, &want) == 0) { g_pin_active = true; g_pin_tid = (pid_t)syscall(SYS_gettid); }
}
static void unpin_last_engine() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (--g_pin_engines > 0) return;
    g_pin_engines = 0;
    if (g_pin_active) { (void)sched_setaffinity(g_pin_tid, sizeof g_pin_old, &g_pin_old); g_pin_active = false; }      // (ESRCH if that thread is gone: nothing to restore)
}

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
static inline int onepass_folders(const pbwtamd_engine *e);

extern "C" void pbwtamd_engine_destroy(pbwtamd_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->s2) (void)hipStreamSynchronize(e->s2);
    if (e->op_prof) {                                       // PBWTAMD_ONEPASS_PROF=1: the last launch's stamps, per tile, relative to the first tile's entry (us)
        std::vector<unsigned long long> hp((size_t)(e->Wt + 64) * 8);    // [tile][8], then [folder][8]
        if (hipMemcpy(hp.data(), e->op_prof, hp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ULL; for (int w = 0; w < e->Wt; ++w) if (hp[(size_t)w * 8]) t0 = std::min(t0, hp[(size_t)w * 8]);
            const char *nm[7] = {"entry", "row ready", "level 1 folded", "level 2 folded", "scattered", "sparse table built", "ranks, range maxima"};
            for (int i : {0, 1, 5, 6, 2, 3, 4}) {
                double mn = 1e30, mx = 0, sum = 0; int n = 0;
                for (int w = 0; w < e->Wt; ++w) { const unsigned long long v = hp[(size_t)w * 8 + i]; if (!v || !hp[(size_t)w * 8]) continue; const double us = (double)(v - t0) * 0.01; mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++n; }
                fprintf(stderr, "[onepass prof] W %d g1 %d %-16s min %.2f mean %.2f max %.2f us after the first tile's entry (%d tiles)\n", e->Wt, e->op_g1, nm[i], mn, n ? sum / n : 0.0, mx, n);
            }
            if (env_int("PBWTAMD_ONEPASS_PROF", 0) > 1)         // = 2: every folder's line (entry, aggregate out) ...
                for (int f = 0; f < 64 && f < onepass_folders(e); ++f)
                    fprintf(stderr, "[onepass folder] %3d %6.2f %6.2f\n", f, (double)(hp[(size_t)(e->Wt + f) * 8] - t0) * 0.01, (double)(hp[(size_t)(e->Wt + f) * 8 + 1] - t0) * 0.01);
            if (env_int("PBWTAMD_ONEPASS_PROF", 0) > 1)         // the scanner form: every scanner's line (entry, local prefixes + aggregate out, exclusive aggregate out)
                for (int f = 0; f < 64 && f < e->op_nscan; ++f)
                    { const unsigned long long *q = &hp[(size_t)(e->Wt + f) * 8];
                      fprintf(stderr, "[onepass scanner] %3d entry %6.2f first watch %6.2f passes done %6.2f aggregate out %6.2f (wave 3 %6.2f) repasses (all launches) %llu | aggregator wave %d done %6.2f\n", f, (double)(q[0] - t0) * 0.01, (double)(q[2] - t0) * 0.01,
                              (double)(q[6] - t0) * 0.01, (double)(q[1] - t0) * 0.01, (double)(q[4] - t0) * 0.01, q[3], f, (double)(q[7] - t0) * 0.01); }
            if (env_int("PBWTAMD_ONEPASS_PROF", 0) > 1)         // ... and every tile's (entry, row, sparse table, ranks, level 1, level 2, scattered)
                for (int w = 0; w < e->Wt; ++w) {
                    fprintf(stderr, "[onepass tile] %4d", w);
                    for (int i : {0, 1, 5, 6, 2, 3, 4}) fprintf(stderr, " %6.2f", hp[(size_t)w * 8 + i] ? (double)(hp[(size_t)w * 8 + i] - t0) * 0.01 : -1.0);
                    fprintf(stderr, "\n");
                }
        }
    }
    shard_release(e);
    if (e->s2) (void)hipStreamDestroy(e->s2);
    if (e->s2_single) (void)hipStreamDestroy(e->s2_single);
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
    if (e->teamctl) (void)dev_free(e->teamctl);
    if (e->teamprof) (void)dev_free(e->teamprof);
    if (e->op_rows) (void)dev_free(e->op_rows);
    if (e->op_grows) (void)dev_free(e->op_grows);
    if (e->op_scanl) (void)dev_free(e->op_scanl);
    for (int i = 0; i < 2; ++i) if (e->h_stage[i]) (void)hipHostFree(e->h_stage[i]);
    if (e->team_host) (void)hipHostFree(e->team_host);
    for (int i = 0; i < 2; ++i) if (e->evD2H[i]) (void)hipEventDestroy(e->evD2H[i]);
    if (e->h2d_stream) { (void)hipStreamSynchronize(e->h2d_stream); (void)hipStreamDestroy(e->h2d_stream); }
    for (int i = 0; i < 2; ++i) if (e->evPrep[i]) (void)hipEventDestroy(e->evPrep[i]);
    for (int i = 0; i < 2; ++i) if (e->evCopy[i]) (void)hipEventDestroy(e->evCopy[i]);
    if (e->op_prof) (void)dev_free(e->op_prof);
    if (e->h_used) (void)hipHostFree(e->h_used);
    if (e->h_nflag) (void)hipHostFree(e->h_nflag);
    if (e->evFlag) (void)hipEventDestroy(e->evFlag);
    for (int i = 0; i < 2; ++i) { if (e->evChain[i]) (void)hipEventDestroy(e->evChain[i]); if (e->evCons[i]) (void)hipEventDestroy(e->evCons[i]); if (e->evRounds[i]) (void)hipEventDestroy(e->evRounds[i]); }
    for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto &p : e->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    void *ptrs[] = {e->A, e->D, e->summ, e->ctl, (void *)e->ctlblk, (void *)e->prof, (void *)e->zerocol, (void *)e->ystale, (void *)e->xTr[0], (void *)e->xTr[1],
                    (void *)e->keysR[0], (void *)e->keysR[1], (void *)e->saveR[0], (void *)e->saveR[1], (void *)e->fillGB[0], (void *)e->fillGB[1], (void *)e->p16r,
                    (void *)e->wflags, (void *)e->nflag, (void *)e->rankdirS, (void *)e->skT, (void *)e->k2agg, (void *)e->k2cnt, e->cols_stage, e->ycols, e->colBytes, (void *)e->p3regs,
                    e->blockCount, e->scal, e->hist, e->hist_rep, e->csum, e->recs, e->yz};
    for (void *p : ptrs) if (p) (void)dev_free(p);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    const bool pinned = e->pinned;
    delete e;
    if (pinned) unpin_last_engine();
}

// set by a caller inside the library around pbwtamd_engine_create: the engine will run the persistent small-panel chain (skel_persist_kernel: the query cursor
// of the query sweeps), not the one-launch round — its tile geometry follows the two-launch round's rule
static thread_local bool g_create_persist = false;
extern "C" int pbwtamd_engine_create(pbwtamd_engine **out, int device, int M, int batch_sites, void *stream) {
    *out = nullptr;
    if (M < 1) return fail("pbwtamd_engine_create: M=%d", M);
    pin_first_engine(device);                               // (before the first HIP call of this library: the runtime's threads inherit the mask)
    struct PinGuard { bool keep = false; ~PinGuard() { if (!keep) unpin_last_engine(); } } pinGuard;      // a create that fails gives its count back
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
    if (e->skel) {                                          // one tiny launch on the chain's stream (a caller's stream may carry a CU mask)
        unsigned *pr = nullptr, hp[65];
        if (dev_alloc((void **)&pr, sizeof hp) == hipSuccess && hipMemsetAsync(pr, 0, sizeof hp, e->stream) == hipSuccess) {
            hipLaunchKernelGGL(xcd_probe_kernel, dim3(64), dim3(64), 0, e->stream, pr);
            if (hipStreamSynchronize(e->stream) == hipSuccess && hipMemcpy(hp, pr, sizeof hp, hipMemcpyDeviceToHost) == hipSuccess) {
                e->xcd_mask = hp[0];
                bool rr = __builtin_popcount(hp[0]) == 8;
                for (int b = 0; rr && b < 64; ++b) rr = hp[1 + b] == hp[1 + (b & 7)];
                for (int b = 0; rr && b < 8; ++b) for (int c = 0; c < b; ++c) if (hp[1 + b] == hp[1 + c]) rr = false;
                e->xcd_rr = rr;
            }
        }
        (void)hipGetLastError();
        if (pr) (void)dev_free(pr);
    }
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
        // (r5r3, with the 256-position tile's row taken from its own chunk tables: 1.004 / 1.040 at 50 k, 1.054 / 1.065 at 60 k, 1.120 / 1.089 at 70 k: up to 256 tiles)
        // under the one-launch round (below) 256-position tiles stay ahead up to ~55 k haplotypes (with the bench consumers and folder workgroups: 0.935 against 0.976
        // us/site at 20 k, 0.987 / 1.026 at 40 k, 1.042 / 1.044 at 50 k; 1.089 / 1.068 at 60 k, 1.193 / 1.106 at 70 k, where 512-position tiles poll both look-back
        // levels at once — profiles/r05_onepass.txt, r5r2), and the 8 193-12 288 exception of the two-launch round goes
        const bool want_onepass = env_int("PBWTAMD_ONEPASS", 1) != 0 && !g_create_persist;
        if (want_onepass && (M + 255) / 256 <= std::min(1024, env_int("PBWTAMD_ONEPASS_MAXW", 320))) e->skEPT = (M <= 65536) ? 1 : 2;
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
            // the one-launch round (pbwt_k_chain.h, skel_onepass_kernel): every tile of a launch must be able to become resident (a tile waits for the rows of tiles
            // before it), so at most 1024 tiles and no more than the device holds at once; it needs neither pair rows nor the two-level scan
            // Measured (profiles/r05_onepass.txt, with the bench consumers): 1.02 against 1.39 us/site at 30 k haplotypes, 1.13 / 1.34 at 50 k, 1.37-1.43 / 1.56-1.68 at 100 k,
            // 1.69 / 1.83 at 150 k; 2.04 / 1.96 at 200 k, 2.45 / 2.12 at 250 k, 5.05 / 3.04 at 500 k (more tiles: longer look-backs, five workgroups per CU): on up to
            // 320 tiles (163 840 haplotypes).  PBWTAMD_ONEPASS=0: the three- / two-launch round; PBWTAMD_ONEPASS_MAXW=n: up to n <= 1024 tiles (tests)
            e->onepass = want_onepass && e->skEPT <= 2 && e->Wt <= std::min(1024, env_int("PBWTAMD_ONEPASS_MAXW", 320));
            // (round 6) wider panels: the SCANNER form (pbwt_k_chain.h) — the scan over the tiles by scanner / aggregator workgroups inside the launch, XCD-local groups,
            // tiles in dispatch order.  Bit-exact at every width (the parity tests run it), and MEASURED SLOWER than three launches per round at every width above 320 tiles
            // (us per round alone: 300 k 18.6 against 16.6, 1 M 38.0 against 26.1; with the bench consumers 2.85 / 2.53 and 6.5 / 4.7 us/site — profiles/r06_onepass.txt has
            // the timeline and the five forms tried): OFF unless PBWTAMD_ONEPASS_SCAN=1.  Counts are 21-bit fields: below 2^21 haplotypes.  PBWTAMD_ONEPASS_SCAN_MIN=n: from
            // n + 1 tiles on (tests: 0)
            const int scan_min = env_int("PBWTAMD_ONEPASS_SCAN_MIN", std::min(1024, env_int("PBWTAMD_ONEPASS_MAXW", 320)));
            e->op_scan = want_onepass && e->skEPT <= 2 && M < (1 << 21) && e->Wt > scan_min && env_int("PBWTAMD_ONEPASS_SCAN", 0) != 0 && e->xcd_rr;
            if (e->op_scan) {
                e->op_g1 = std::max(4, std::min(32, env_int("PBWTAMD_ONEPASS_SCAN_G", 32)));
                e->op_nscan = (e->Wt + e->op_g1 - 1) / e->op_g1;
                int per_cu = 0, ncu = 0;
                const hipError_t r1 = (e->skEPT == 1) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skel_onepass_scan_kernel<1>, BLOCK, 0)
                                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skel_onepass_scan_kernel<2>, BLOCK, 0);
                if (r1 != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
                if ((long long)std::min(per_cu, 6) * ncu < (long long)sk1_front(e->op_nscan) + 2 * 8 * e->op_g1 || e->op_nscan > 128) e->op_scan = false;      // the front and two runs of tiles must fit beside each other      // the scanners and two groups of tiles must fit beside each other
                e->onepass = e->op_scan;
            } else
            if (e->onepass) {
                int per_cu = 0, ncu = 0;
                const hipError_t r1 = (e->skEPT == 1) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skel_onepass_kernel<1>, BLOCK, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skel_onepass_kernel<2>, BLOCK, 0);
                if (r1 != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
                const long long cap = (long long)std::min(per_cu, 6) * ncu;                  // (6: what the hardware admits of a kernel with ~100 SGPRs whatever the API says)
                e->op_cap = cap;
                if (cap < e->Wt) e->onepass = false;
                // a folder workgroup per group of ceil(sqrt(W)) tiles behind the tiles, where they too fit: -5 % alone / -2.6 % beside the consumers at 100 k, -6 / -4 % at
                // 50 k, -3 / -2 % at 130 k (254 tiles); +3.5 % beside the consumers at 150 k (294 tiles): up to 256 tiles (profiles/r05_onepass.txt, r5i)
                int gg = 1; while (gg * gg < e->Wt) ++gg;
                e->op_fk = std::max(1, std::min(4, env_int("PBWTAMD_ONEPASS_FOLDERS_K", 1)));
                e->op_folders = e->onepass && env_int("PBWTAMD_ONEPASS_FOLDERS", 1) != 0 && e->Wt <= env_int("PBWTAMD_ONEPASS_FOLDERS_MAXW", 256) && cap >= e->Wt + e->op_fk * gg;
                e->op_merged = env_int("PBWTAMD_ONEPASS_MERGED", e->skEPT == 1 ? 1 : 0) != 0;
            }
            e->op_both = e->op_folders && env_int("PBWTAMD_ONEPASS_BOTH", e->skEPT == 2 ? 1 : 0) != 0;
            e->op_ordered = env_int("PBWTAMD_ONEPASS_ORDERED", 0) != 0;
            if (e->onepass && !e->op_scan) { e->op_g1 = 1; while (e->op_g1 * e->op_g1 < e->Wt) ++e->op_g1; if (const char *sg = tune_env("PBWTAMD_ONEPASS_G1")) e->op_g1 = std::max(2, std::min(atoi(sg), 64)); }     // groups of ceil(sqrt(W)) tiles: as many groups as tiles per group
            e->W2 = (e->Wt + 1) / 2;
            static const int prow_min = tune_env("PBWTAMD_PROW_MIN") ? atoi(tune_env("PBWTAMD_PROW_MIN")) : 136;
            static const bool prow_ept1 = tune_env("PBWTAMD_PROW_EPT1") && atoi(tune_env("PBWTAMD_PROW_EPT1"));   // measurement builds: pairs of 256-position tiles
            e->prow = !e->onepass && pair_rows && (e->skEPT == 2 || (e->skEPT == 1 && prow_ept1)) && e->W2 > prow_min && e->W2 <= prow_max;
            if (e->skEPT == 2 && e->Wt > 2048 && !e->prow) { const int r = fail("pbwtamd_engine_create: %d tiles of 512 positions need pair rows", e->Wt); pbwtamd_engine_destroy(e); return r; }
            e->strideS = e->prow ? (size_t)SKK * e->W2 * 2 + SKK / 2 : (size_t)SKK * e->Wt + SKK / 2;
            // wide panels (more than 512 scan rows): the scan in its local form (skel_k2_local_kernel) — one exclusive aggregate row per scan workgroup
            // (<= 64 of them) behind the round's other tables; PBWTAMD_K2_LOCAL=0: the two-pass form (skel_k2_wide_kernel)
            {
                const int rows = e->prow ? e->W2 : e->Wt;
                // (interleaved A/B, end to end: -5.4 % at 600 k haplotypes, -2.6 % at 1 M — the chain alone 3.53 -> 3.27 us/site; +3 % at 2 M, where 62 scan
                // workgroups make the last arriver's fold long and the consumers, not the chain, set the pace: up to 1024 rows)
                // Below 513 rows the one-level scan (skel_k2_kernel) stays ahead: 2.88 against 3.15 us/site at 500 k, 2.48 / 2.74 at 350 k, 1.90 / 2.28 at 200 k
                // (PBWTAMD_K2_LOCAL_MIN=n, A/B and parity runs only: the local form from n + 1 rows on).
                e->k2local = !e->onepass && rows > env_int("PBWTAMD_K2_LOCAL_MIN", 512) && rows <= env_int("PBWTAMD_K2_LOCAL_MAX", 1024) && env_int("PBWTAMD_K2_LOCAL", 1) != 0;
                // (16 rows per scan workgroup, twice the arrivals: 4.70 against 4.57 us/site at 1 M, 3.60 / 3.53 at 600 k; PBWTAMD_K2_LOCAL_MAX=2048, A/B: 64-row
                // workgroups there — 6.34 against 6.36 us/site at 1.5 M, 8.20 against 8.04 at 2 M: not taken)
                e->k2tpw = rows > 1024 ? 64 : 32;
                // the local form's capacity: aggx and k2agg hold 64 rows (one per scan workgroup), so rows <= 64 * k2tpw — odd PBWTAMD_K2_LOCAL_MIN / _MAX
                // combinations fall back to the other scans instead of writing past them
                if ((rows + e->k2tpw - 1) / e->k2tpw > 64) e->k2local = false;
                e->aggx_off = e->k2local ? e->strideS : 0;
                if (e->k2local) e->strideS += (size_t)64 * SKK;
            }
            for (int i = 0; i < 2; ++i) {
                ALLOC(e->keysR[i], (size_t)(rounds + 1) * e->Mpad);
                ALLOC(e->saveR[i], (size_t)rounds * e->strideS * sizeof(int2));
                ALLOC(e->fillGB[i], (size_t)rounds * SKK * sizeof(int2));
            }
            // (the 16-bit hand-off ring, 2 x (B + 2) slots of strideD 16-bit words — 2 GB at 1 M haplotypes — is allocated by run_consumers with the first
            // batch that takes the packed path: query-sweep engines, record / checksum option sets, sharded ranks and the P engines of
            // pbwtamd_pass_advance_many that never take it do not pay for it)
        }
        ALLOC(e->skT, (size_t)(e->Wt + 1) * SKK * sizeof(int2));
        if (e->onepass) {
            const int ngrp = (e->op_scan ? 2 : 1) * ((e->Wt + e->op_g1 - 1) / e->op_g1);        // (scanner form: the groups' aggregates, then their exclusive folds)
            ALLOC(e->op_rows, (size_t)(e->Wt + 64) * SKK * sizeof(unsigned long long));       // (+ 64 rows: the scanner form's windows read ahead of a group's last row)
            ALLOC(e->op_grows, (size_t)ngrp * SKK * sizeof(unsigned long long));
            ECHK(hipMemsetAsync(e->op_rows, 0, (size_t)e->Wt * SKK * sizeof(unsigned long long), e->stream));        // tag 0: no launch has published yet (the first launch's tag is 1)
            ECHK(hipMemsetAsync(e->op_grows, 0, (size_t)ngrp * SKK * sizeof(unsigned long long), e->stream));
            if (e->op_scan) {
                ALLOC(e->op_scanl, (size_t)e->Wt * SKK * sizeof(unsigned long long));
                ECHK(hipMemsetAsync(e->op_scanl, 0, (size_t)e->Wt * SKK * sizeof(unsigned long long), e->stream));
            }
            if (env_int("PBWTAMD_ONEPASS_PROF", 0)) { ALLOC(e->op_prof, (size_t)(e->Wt + 64) * 8 * sizeof(unsigned long long)); ECHK(hipMemsetAsync(e->op_prof, 0, (size_t)(e->Wt + 64) * 8 * sizeof(unsigned long long), e->stream)); }
        }
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
    e->pinned = true; pinGuard.keep = true;                 // from here on pbwtamd_engine_destroy gives the count back
    *out = e;
    return 0;
}

static int flush_pending(pbwtamd_engine *e);
static int d2h_staged(pbwtamd_engine *e, void *dst, const void *src, size_t bytes, hipStream_t st);

extern "C" int pbwtamd_sync(pbwtamd_engine *e) {
    HIPCHK(hipSetDevice(e->device));
    CHK(flush_pending(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->s2));
    int err = 0;
    HIPCHK(hipMemcpy(&err, e->ctl + 2, sizeof(int), hipMemcpyDeviceToHost));
    if (err) return fail("pbwtamd: device-side error flag %d (1=histogram range, 2/3=malformed packed column, 4=yz buffer overflow, 5=tile scan of a wide panel timed out "
                         "waiting for its workgroups, 6/7=a position-sharded rank timed out waiting for its peers, 9=a peer reported its own failure, 10=a team of the "
                         "team-persistent chain did not fill, 11=a tile of the one-launch round timed out waiting for the tiles before it)", err);
    return 0;
}

#include "pbwt_e_pass.inc"
#include "pbwt_e_consumers.inc"
#include "pbwt_e_chain.inc"
#include "pbwt_e_host.inc"
#include "pbwt_e_query.inc"
