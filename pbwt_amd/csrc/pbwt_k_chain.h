// pbwt_k_chain.h — the SKELETON chain: transpose32, skel_hist / skel_k2 / skel_k2_wide / skel_rank (8 sites per round), the position-sharded forms, the persistent and many-panel forms, read-side keys.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// =============================================================================================
// SKELETON + FILL.  The critical chain advances EIGHT sites per round with three
// launches (K1, K2, K3) and produces only every 8th state; the seven states in between are filled
// in afterwards by batched single-site kernels that run over all blocks of a batch at once.
//   a_{k+8} = stable sort of a_k by the 8-bit key (bit j = allele at site k+j);
//   d_{k+8}[e] = range max of d_k since the previous element with the same key (level-0 order), or
//                k+1+msb(key ^ key') with key' the nearest lower non-empty key when there is none
//   (tests/tile_model.py::stepB_tiles).  Tiles of 1024 positions, 256 threads.
// ---------------------------------------------------------------------------------------------
constexpr int SKB = 8, SKK = 1 << SKB;

// 32 sites x 32 haplotypes bit transpose: xT[blk][h] bit j = allele of haplotype h at site 32*blk + j
// (sites at or beyond n_valid read as 0).  grid (ceil(wpc/256), nblk).
__global__ __launch_bounds__(BLOCK) void transpose32_kernel(const uint32_t *cols, int wpc, int n_valid, uint32_t *xT, size_t strideX, int Mpad) {
    const int wd = blockIdx.x * BLOCK + threadIdx.x, blk = blockIdx.y;
    if (wd >= wpc) return;
    uint32_t r[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { const int site = blk * 32 + j; r[j] = (site < n_valid) ? cols[(size_t)site * wpc + wd] : 0u; }
    // r[j] bit i = hap 32*wd+i at site j  ->  r[i] bit j: five butterfly stages (80 swaps instead of 1024 bit moves)
#pragma unroll
    for (int j = 16, st = 0; st < 5; ++st, j >>= 1) {
        const uint32_t m = (j == 16) ? 0x0000ffffu : (j == 8) ? 0x00ff00ffu : (j == 4) ? 0x0f0f0f0fu : (j == 2) ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k & j) continue;                            // pairs (k, k + j) with bit j of k clear
            const uint32_t tt = ((r[k] >> j) ^ r[k + j]) & m;
            r[k + j] ^= tt; r[k] ^= tt << j;
        }
    }
    // four BYTE planes per 32-site block: plane 4 blk + q holds, per haplotype, the alleles of sites 32 blk + 8 q .. + 7 = the 8-bit
    // key of one radix step.  A round gathers its next keys from ONE plane: Mpad bytes (1 MB at M = 1 M, L2-resident) instead of
    // 4-byte words of a 4 MB array — the rank kernel's gather was 42 of its 61 MB of HBM-side traffic per launch at that width.
    if (wd * 32 >= Mpad) return;
    unsigned char *base = reinterpret_cast<unsigned char *>(xT) + (size_t)blk * 4 * strideX + (size_t)wd * 32;   // strideX = Mpad: bytes per plane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            pk[j] = ((r[4 * j] >> (8 * q)) & 0xffu) | (((r[4 * j + 1] >> (8 * q)) & 0xffu) << 8) | (((r[4 * j + 2] >> (8 * q)) & 0xffu) << 16) | (((r[4 * j + 3] >> (8 * q)) & 0xffu) << 24);
        uint4 *dst = reinterpret_cast<uint4 *>(base + (size_t)q * strideX);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
}

// HIST (K1): per tile of T = 256*EPT positions — count the 8-bit keys, and the max of d_k after each
// key's last occurrence (whole-tile max for absent keys).  The keys travel with the state (the rank
// kernel of the previous round scattered them), so this reads 1 + 4 bytes per position.  Threads own
// positions in REVERSE blocked order so that a forward scan over threads is a suffix scan over
// positions.  Output: tbl[tile][key] {count, tail}.
// Workgroups are dealt to the 8 XCDs round-robin by linear id (observed: block b runs on XCD b % 8) and every XCD has its own
// L2.  xcd_tile gives XCD x a CONTIGUOUS range of logical tiles, so that neighbouring tiles — which write neighbouring
// destinations of the same bucket — complete their 64-B lines in one L2 instead of eight.  Placement is for speed only.
__device__ __forceinline__ int xcd_tile(int lin, int n) {
    const int q = n >> 3, r = n & 7, x = lin & 7, idx = lin >> 3;
    return x * q + min(x, r) + idx;
}

struct SkArgs {
    const int *a; const int *d; const unsigned char *keys;     // input state and its 8-bit keys
    int *a_out; int *d_out; unsigned char *keys_out;
    int2 *tbl;                                                  // hist -> scan: [W][256] {count, tail}
    int2 *tbl0;                                                 // pair rows: the same pair for the FIRST HALF of every hist tile (kept beside the scan)
    int pair;                                                   // rank: the scan rows are per PAIR of tiles (row w / 2); odd tiles fold tbl0[w / 2] in
    int2 *scan; int *total;                                     // scan -> rank (kept for the fill): [W][256] {keys before the tile, carry}, total[256]
    const unsigned char *kbnext; int has_next;                  // byte plane of the NEXT round's keys by haplotype (transpose32_kernel)
    const unsigned long long *ycnext;                           // read side: sorted bit column of the OUTPUT state's site (tag by position); keys are precomputed
    int M, W, k;                                                // k = site of the input state; W = tiles of this launch
    int xcd;                                                    // bit 1: rank, bit 2: hist — XCD-contiguous tiles (xcd_tile)
    int w0, Wtot;                                               // position sharding: this launch covers tiles w0 .. w0+W-1 of Wtot (one GPU: 0, W)
    // (round 4) wide panels, skel_k2_local_kernel: scan[] holds prefixes LOCAL to the scan workgroup of aggx_tpw rows, and aggx[row / aggx_tpw][key] the
    // exclusive fold of the workgroups before it; whoever reads a row folds the two (sk_fold_aggx).  nullptr: scan[] holds the global prefixes.
    const int2 *aggx = nullptr; int aggx_tpw = 0;
    // (round 5) the ONE-LAUNCH round (skel_onepass_kernel): rows[tile][key] / grows[group][key] = tagged {count, tail} granules published by the tiles / by the
    // last tile of every group of g1 consecutive tiles; tag = this launch's epoch (11 bits); total[] is precomputed (skel_totals_kernel)
    unsigned long long *rows = nullptr, *grows = nullptr; int g1 = 0; unsigned tag = 0; int *err = nullptr;
    int nfold = 0;                                              // > 0: the launch carries one FOLDER workgroup per group behind its W tiles (it folds the group's rows into the aggregate); 0: the group's last tile does
    unsigned long long *prof = nullptr;                         // PBWTAMD_ONEPASS_PROF=1: [tile][8] wall-clock stamps of the launch (debugging aid)
    // (round 6) the one-launch round's SCANNER form (wide panels): nscan scanner workgroups IN FRONT of the W tiles; scanner s turns the rows of tiles
    // s * g1 .. into scanl[tile][key] (the prefix local to the group) and grows[s] (the group's aggregate), then grows[nscan + s] (the fold of the groups before)
    unsigned long long *scanl = nullptr; int nscan = 0;
};
// the global (keys before, carry) of a row from the aggregate of the scan workgroups before its own (L) and its prefix local to that workgroup (R):
// skel_k2_kernel's combine, with the carry of "no earlier occurrence" = -1 on the way out
__device__ __forceinline__ int2 sk_fold_aggx(int2 L, int2 R) {
    return make_int2(L.x + R.x, R.x ? R.y : (L.x ? max(L.y, R.y) : -1));
}

// How a chain kernel reads what ANOTHER workgroup of the SAME launch has written (the persistent forms below; DESIGN.md section 2, profiles/r04_latprobe4.txt):
//   SKM_LAUNCH  the three-launch round: producer and consumer are different launches, plain loads;
//   SKM_TEAM    a team of workgroups on ONE XCD (skel_team_kernel): producers store plainly (the data stays in that XCD's L2, the team's coherence
//               point), consumers bypass their CU's L1 with nontemporal loads — plain loads hit stale L1 lines (37-43 % in the probe);
//   SKM_SPREAD  teams on several XCDs: write-through (sc1) stores and sc1 loads on both sides (agent-scope relaxed atomics).
constexpr int SKM_LAUNCH = 0, SKM_TEAM = 1, SKM_SPREAD = 2;
template <int MODE, typename T>
__device__ __forceinline__ T skm_ld(const T *p) {
    if constexpr (MODE == SKM_LAUNCH) return *p;
#ifndef SKM_TEAM_SC1
    else if constexpr (MODE == SKM_TEAM) return __builtin_nontemporal_load(p);
#endif
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ int2 skm_ld2(const void *p) {      // 8 bytes, 8-byte aligned
    if constexpr (MODE == SKM_LAUNCH) return *reinterpret_cast<const int2 *>(p);
    else { const unsigned long long v = skm_ld<MODE>(reinterpret_cast<const unsigned long long *>(p)); return make_int2((int)(unsigned)v, (int)(v >> 32)); }
}
template <int MODE, typename T>
__device__ __forceinline__ void skm_st(T *p, T v) {
    if constexpr (MODE == SKM_SPREAD) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int MODE>
__device__ __forceinline__ void skm_st2(int2 *p, int2 v) {
    if constexpr (MODE == SKM_SPREAD) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Position sharding: the ranks of one panel own contiguous ranges of TILES of the sorted order.  Every rank keeps
// full-width ring slots; the chain of rank g reads and writes positions pb[g] .. pb[g+1]-1 of them only, and its rank kernel
// stores each (a | tag, d', key) into the slot of the position's OWNER through the peers' mapped ring pointers (hipIpc).
constexpr int SHARD_MAX = 8;
struct SkShardOut {
    int n;                                                      // ranks
    int pb[SHARD_MAX + 1];                                      // first position of every rank's range; pb[n] = M (unused entries: INT_MAX)
    int *a[SHARD_MAX]; int *d[SHARD_MAX]; unsigned char *k[SHARD_MAX];   // the OUTPUT slot (and its key row) in every rank's ring
    const int *err;                                             // the engine's error word: once set, the rank kernel scatters nothing
};

// HALF (pair rows, wide panels): the workgroup covers a PAIR of the rank kernel's tiles and also emits the (count, tail) row of its
// first half; the scan over the tiles then runs on half as many rows (the scan launch is what a wide panel pays most for beside
// the consumers: 22.7 us per round at 1954 rows, 14 at 977), and the rank / fill workgroup of an odd tile folds the first half's
// row into its pair's prefix (skel_k2_kernel's combine).  Waves 2, 3 hold the first half.
// skel_hist_row: the row of tile w for key t = threadIdx.x — {count, tail} (and {c0, tl0} of the first half with HALF) — left in registers; skel_hist_body stores it
template <int EPT, bool HALF, int MODE = SKM_LAUNCH>
__device__ __forceinline__ void skel_hist_row(const SkArgs &g, int w, int &c_out, int &tl_out, int &c0_out, int &tl0_out) {
    static_assert(MODE == SKM_LAUNCH || EPT <= 2, "the persistent forms run 256- and 512-position tiles");
    constexpr int T = BLOCK * EPT;
    __shared__ int h_cnt[SKK], h_last[SKK];
    __shared__ int s_suf[T];
    __shared__ int s_w[WAVES];
    __shared__ int h_cnt0[HALF ? SKK : 1], h_last0[HALF ? SKK : 1], s_suf0[HALF ? T / 2 : 1];
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int rb = BLOCK - 1 - t;
    const int l0 = rb * EPT, i0 = w * T + l0;
    unsigned packed;
    int dv[EPT];
    if constexpr (EPT == 4) {
        packed = *reinterpret_cast<const unsigned *>(g.keys + i0);
        const int4 vd = *reinterpret_cast<const int4 *>(g.d + i0);
        dv[0] = vd.x; dv[1] = vd.y; dv[2] = vd.z; dv[3] = vd.w;
    } else if constexpr (EPT == 2) {
        packed = skm_ld<MODE>(reinterpret_cast<const unsigned short *>(g.keys + i0));
        const int2 vd = skm_ld2<MODE>(g.d + i0);
        dv[0] = vd.x; dv[1] = vd.y;
    } else {
        packed = skm_ld<MODE>(g.keys + i0); dv[0] = skm_ld<MODE>(g.d + i0);
    }
    h_cnt[t] = 0; h_last[t] = -1;
    if (HALF) { h_cnt0[t] = 0; h_last0[t] = -1; }
    int key[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool valid = i0 + e < g.M;
        key[e] = valid ? (int)((packed >> (8 * e)) & 0xffu) : -1;
        if (!valid) dv[e] = 0;
    }
    lds_barrier();
    // one LDS atomic pair per (wave, key) instead of per position: real panels are skewed (most positions share the all-zero
    // key), and same-address LDS atomics serialise.  The first lane of a key group holds its highest position (reverse order).
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        unsigned long long same = __ballot(key[e] >= 0);
#pragma unroll
        for (int b = 0; b < SKB; ++b) { const unsigned long long bal = __ballot((key[e] >> b) & 1); same &= ((key[e] >> b) & 1) ? bal : ~bal; }
        if (key[e] >= 0 && (same & ((lane == 0) ? 0ULL : (~0ULL >> (64 - lane)))) == 0) {
            atomicAdd(&h_cnt[key[e]], __popcll(same)); atomicMax(&h_last[key[e]], l0 + e);
            if (HALF && wv >= 2) { atomicAdd(&h_cnt0[key[e]], __popcll(same)); atomicMax(&h_last0[key[e]], l0 + e); }
        }
    }
    int own = dv[0];
#pragma unroll
    for (int e = 1; e < EPT; ++e) own = max(own, dv[e]);
    int inc = wave_iscan_max(own);                         // lanes before me = positions after mine
    if (lane == 63) s_w[wv] = inc;
    const int excl_lane = lane_shr1(inc, 0);
    lds_barrier();
    int later = excl_lane, later0 = excl_lane;
    for (int q = 0; q < wv; ++q) later = max(later, s_w[q]);
    if (HALF && wv == 3) later0 = max(later0, s_w[2]);
#pragma unroll
    for (int e = EPT - 1; e >= 0; --e) {                   // s_suf[l] = max d over positions > l (s_suf0: inside the first half)
        s_suf[l0 + e] = later; later = max(later, dv[e]);
        if (HALF && wv >= 2) { s_suf0[l0 + e] = later0; later0 = max(later0, dv[e]); }
    }
    int tilemax = 0;
    for (int q = 0; q < WAVES; ++q) tilemax = max(tilemax, s_w[q]);
    lds_barrier();
    const int c = h_cnt[t];
    c_out = c; tl_out = c ? s_suf[h_last[t]] : tilemax;
    if (HALF) { const int c0 = h_cnt0[t]; c0_out = c0; tl0_out = c0 ? s_suf0[h_last0[t]] : max(s_w[2], s_w[3]); }
}
template <int EPT, bool HALF, int MODE = SKM_LAUNCH>
__device__ __forceinline__ void skel_hist_body(const SkArgs &g, int wsel = -1) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);                          // the dependent chain shares SIMDs with the throughput kernels of the consumer stream: issue first
#endif
    const int t = threadIdx.x, w = (wsel >= 0) ? wsel : g.w0 + ((g.xcd & 4) ? xcd_tile(blockIdx.x, g.W) : blockIdx.x);
    int c = 0, tl = 0, c0 = 0, tl0 = 0;
    skel_hist_row<EPT, HALF, MODE>(g, w, c, tl, c0, tl0);
    skm_st2<MODE>(g.tbl + (size_t)w * SKK + t, make_int2(c, tl));      // row-major: one coalesced 2 KB row per tile
    if (HALF) g.tbl0[(size_t)w * SKK + t] = make_int2(c0, tl0);
}
template <int EPT, bool HALF = false>
__global__ __launch_bounds__(BLOCK) void skel_hist_kernel(SkArgs g) { skel_hist_body<EPT, HALF>(g); }

// SCAN (K2): exclusive scan over the W tiles, per key, of the pair (count, max d since the key's last
// occurrence) with combine(L,R) = (L.c+R.c, R.c ? R.t : max(L.t,R.t)) (for a tile without the key, t
// is the tile's max).  A workgroup owns KPW keys: it pulls the [W][KPW] slab of the row-major table
// through LDS (8*KPW-byte row segments: 32-byte sectors at KPW = 4), each wave scans KPW/4 keys
// with lanes = tiles (TPL consecutive tiles per lane, DPP scan across lanes), and the slab goes back
// the same way.  Output scan[tile][key] = {keys before the tile, carry (-1: no earlier occurrence)},
// total[key].  grid = 256 / KPW workgroups of KPW waves.
struct Sk2Args { const int2 *tbl; int2 *scan; int *total; int W; };
template <int KPW, int TPL, int MODE = SKM_LAUNCH>
__device__ __forceinline__ void skel_k2_body(const Sk2Args &g, int kblk = -1) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int NT = KPW * 64;                            // one wave per key
    constexpr int WP = 64 * (TPL + 1);                      // a lane's TPL tiles + one pad entry: lane stride TPL+1 is odd, no LDS bank conflicts
    __shared__ int2 s_v[KPW][WP];
    const int t = threadIdx.x, lane = lane_id(), kk = t >> 6, key0 = ((kblk >= 0) ? kblk : (int)blockIdx.x) * KPW;
    constexpr int NIT = 64 * TPL * KPW / NT;                // = TPL: all loads in flight at once (one round trip, not NIT)
    int2 ld[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * NT, r = idx / KPW, kq = idx % KPW;
        ld[i] = (r < g.W) ? skm_ld2<MODE>(g.tbl + (size_t)r * SKK + key0 + kq) : make_int2(0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * NT, r = idx / KPW, kq = idx % KPW;
        s_v[kq][r + r / TPL] = ld[i];
    }
    __syncthreads();
    {
        int c[TPL], tt[TPL];
        int sc = 0, st = 0;                                // this lane's tiles combined
#pragma unroll
        for (int x = 0; x < TPL; ++x) {
            const int w = lane * TPL + x;
            const int2 v = (w < g.W) ? s_v[kk][lane * (TPL + 1) + x] : make_int2(0, 0);
            c[x] = v.x; tt[x] = v.y;
            st = c[x] ? tt[x] : max(st, tt[x]); sc += c[x];
        }
        int ic = sc, it = st;                              // inclusive wave scan of (sc, st)
#define SK2_STEP(CTRL, RM) { const int lc = dpp_mov<CTRL, RM>(0, ic), lt2 = dpp_mov<CTRL, RM>(0, it); it = ic ? it : max(lt2, it); /* uses OLD ic = R.c */ ic += lc; }
        SK2_STEP(0x111, 0xf) SK2_STEP(0x112, 0xf) SK2_STEP(0x114, 0xf) SK2_STEP(0x118, 0xf) SK2_STEP(0x142, 0xa) SK2_STEP(0x143, 0xc)
#undef SK2_STEP
        int ec = lane_shr1(ic, 0), et = lane_shr1(it, 0);  // exclusive prefix of this lane's first tile
#pragma unroll
        for (int x = 0; x < TPL; ++x) {
            const int w = lane * TPL + x;
            if (w < g.W) s_v[kk][lane * (TPL + 1) + x] = make_int2(ec, ec ? et : -1);
            et = c[x] ? tt[x] : max(et, tt[x]); ec += c[x];
        }
        if (lane == 63) skm_st<MODE>(g.total + key0 + kk, ic);
    }
    __syncthreads();
    for (int idx = t; idx < g.W * KPW; idx += NT) {
        const int r = idx / KPW, kq = idx % KPW;
        skm_st2<MODE>(g.scan + (size_t)r * SKK + key0 + kq, s_v[kq][r + r / TPL]);
    }
}
template <int KPW, int TPL>
__global__ __launch_bounds__(KPW * 64) void skel_k2_kernel(Sk2Args g) { skel_k2_body<KPW, TPL>(g); }

// SCAN for wide panels (more than 512 tiles): the per-key scan over the tiles in two levels inside ONE launch.
// skel_k2_kernel reads the row-major table in 16-byte pieces of 2 KB rows (a quarter of every 64-byte sector is used) and
// walks 32 tiles per lane serially: 13.9 us at M = 1 M (1954 tiles).  Here a workgroup owns TPW consecutive TILES and all
// 256 keys (thread = key): whole rows, every load in flight at once; it publishes its (count, carry) aggregate per key,
// arrives on a counter, and once all workgroups have arrived folds the aggregates of the workgroups before it.
// All of them are resident at once (W / TPW <= 64 workgroups).  Cross-workgroup visibility: 8-byte agent-scope relaxed
// atomics on both sides (write-through stores, L1-bypassing loads), `s_waitcnt vmcnt(0)` before the arrival — the
// granule form of MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility".
struct Sk2WArgs { const int2 *tbl; int2 *scan; int *total; int W; unsigned long long *agg; unsigned *counter; unsigned target; int *err; int2 *aggx; };

// (round 4) The same scan WITHOUT its second half.  skel_k2_wide_kernel is a chain of eight dependent round trips: two batches of rows, the aggregate out,
// the arrival and the wait for everybody, the error word, the aggregates in, the rows again, the prefixes out — and every one of them stretches beside the
// consumers.  Here a workgroup writes, in its first and only pass over its rows, the prefixes LOCAL to itself (raw: count and running tail, the tail of
// a key not yet seen being the maximum so far), publishes its aggregate and arrives; nobody waits: the workgroup that arrives LAST folds the <= 64
// aggregates into one exclusive row per workgroup (aggx) and the totals.  The rank and fill workgroups fold aggx[row / TPW] in front of their row with
// one more 8-byte load per thread (sk_fold_aggx), issued together with the row's.  Five round trips, the last two in one workgroup only.
template <int TPW, int CH = 16>
__global__ __launch_bounds__(SKK) void skel_k2_local_kernel(Sk2WArgs g) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    __shared__ int s_last;
    const int t = threadIdx.x, j = blockIdx.x, w0 = j * TPW;
    int lc = 0, lt = 0;                                      // running local prefix for key t (d >= 0: 0 is the identity of the tails' maximum)
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (w0 + x0 + x < g.W) ? g.tbl[(size_t)(w0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) {
            if (w0 + x0 + x < g.W) g.scan[(size_t)(w0 + x0 + x) * SKK + t] = make_int2(lc, lt);
            lt = v[x].x ? v[x].y : max(lt, v[x].y); lc += v[x].x;
        }
    }
    __hip_atomic_store(g.agg + (size_t)j * SKK + t, ((unsigned long long)(unsigned)lt << 32) | (unsigned)lc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_last = (__hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == g.target) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const int nwg = (int)gridDim.x;                          // <= 64
    int ec = 0, et = 0;
#pragma unroll 1
    for (int i0 = 0; i0 < nwg; i0 += 32) {
        unsigned long long pv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) pv[i] = (i0 + i < nwg) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i0 + i < nwg) g.aggx[(size_t)(i0 + i) * SKK + t] = make_int2(ec, et);
            const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32);
            et = vc ? vt : max(et, vt); ec += vc;
        }
    }
    g.total[t] = ec;
}
// 16 rows / 32 aggregates in flight per lane, and the rows are read a second time (from L2) for the output pass.  The first
// form of this kernel held all 32 rows + 64 aggregates in 200 VGPRs (one round trip each, 8.0 us alone).  A 200-VGPR wave fits
// on no SIMD while a consumer kernel is at full occupancy (sweep: 8 waves x 56 VGPRs, fill: 6 x 56), and the 56 registers a
// retiring consumer workgroup frees go to the next consumer workgroup: measured (rocprofv3 trace), that launch waited for the
// END of the fill, 1.1-1.4 ms, and the chain stood still beside fill + sweep.  This form (no LDS; 74 VGPRs with this compiler, 44 with 16
// aggregates in flight — measured equal at the end of round 3: 5.94 against 5.91 us/site at 1 M) runs beside the consumers: 9.3 us alone, 14 us
// beside the fill instead of 185; end to end at 1 M 7.25 -> 6.25 us/site.  Around the shipped (rows, aggregates) = (16, 32): (8, 32) 6.21,
// (32, 32) 6.25, (16, 64) 6.93, (32, 64) 7.12 against 6.12.
template <int TPW, int CH = 16, int PCH = 32>                // rows / aggregates in flight per lane
__global__ __launch_bounds__(SKK) void skel_k2_wide_kernel(Sk2WArgs g) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int t = threadIdx.x, j = blockIdx.x, w0 = j * TPW;
    int ac = 0, at = 0;                                      // this workgroup's aggregate for key t
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (w0 + x0 + x < g.W) ? g.tbl[(size_t)(w0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) { at = v[x].x ? v[x].y : max(at, v[x].y); ac += v[x].x; }
    }
    __hip_atomic_store(g.agg + (size_t)j * SKK + t, ((unsigned long long)(unsigned)at << 32) | (unsigned)ac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded wait (~1 s): if an earlier launch of this chain never ran, the arrivals it owes never come — flag it (device
        // error 5, reported at the next pbwtamd_sync) instead of hanging the GPU
        int spins = 0;
        while ((int)(__hip_atomic_load(g.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 25) || ((spins & 4095) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { atomicCAS(g.err, 0, 5); break; }
        }
    }
    __syncthreads();
    if (__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // incomplete aggregates: the batch has failed, no output pass (no barrier follows)
    int ec = 0, et = 0;                                      // prefix over the workgroups before this one
#pragma unroll 1
    for (int i0 = 0; i0 < j; i0 += PCH) {
        unsigned long long pv[PCH];
#pragma unroll
        for (int i = 0; i < PCH; ++i) pv[i] = (i0 + i < j) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
        for (int i = 0; i < PCH; ++i) {
            const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32);     // beyond j: (0, 0), the identity
            et = vc ? vt : max(et, vt); ec += vc;
        }
    }
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {                   // output pass: the rows again (L2), the running prefix written in front of each
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (w0 + x0 + x < g.W) ? g.tbl[(size_t)(w0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) {
            if (w0 + x0 + x < g.W) g.scan[(size_t)(w0 + x0 + x) * SKK + t] = make_int2(ec, ec ? et : -1);
            et = v[x].x ? v[x].y : max(et, v[x].y); ec += v[x].x;
        }
    }
    if (j == (int)gridDim.x - 1) g.total[t] = ec;
}

// ---------------------------------------------------------------------------------------------
// POSITION SHARDING across GPUs (pbwtCore.c:485-508 is what is sharded).  With the skeleton the per-site
// "exclusive scan of local counts + all-to-all" of the north star becomes, per ROUND of 8 sites:
//   (1) every rank publishes ONE row of 256 {count, tail} — its tiles' rows folded with the scan's own combine — into every
//       peer's exchange block, and the scan of a rank starts from the fold of the rows of the ranks before it;
//   (2) the rank kernel stores (a | tag, d', key) straight into the owner's ring slot (peer stores through hipIpc mappings),
//       and a flag barrier closes the round.
// The exchange block lives in device memory of its owner, mapped into every peer; everything in it is accessed with
// system-scope atomics only (no cached copies), the bulk data only across kernel boundaries (tools/ipcprobe.hip measures both).
struct alignas(256) ShardXch {
    unsigned f1[64];                                        // [src] round whose row aggregate src has published here
    unsigned f2[64];                                        // [src] chain barriers src has arrived at (scatter of a round complete)
    unsigned f3[64];                                        // [src] batches whose consumers src has finished (ring reuse)
    unsigned long long ragg[SHARD_MAX][SKK];                // [src][key] count | tail << 32 of src's tiles, current round
    unsigned perr[64];                                      // [src] nonzero: rank src has failed (a bounded wait ran out there) — whoever waits here stops waiting
};
struct ShardPeers { ShardXch *x[SHARD_MAX]; int n, me; };

// Bounded (seconds): a rank that died must not hang the others' GPUs.  The error is STICKY: once the engine's error word is set (here, by
// another workgroup, by an earlier launch) or a peer has flagged itself failed, nobody waits again — the rest of the pass falls through its
// waits, the kernels skip their stores (skel_k2s_kernel, skel_rank_shard_kernel), and the host fails the pass at its next event poll.
// What the waits cost is kept for the host (pbwtamd_shard_stats; err = the engine's ctl + 2, the counters are ctl + 8 ..): code 7 — a round's rows of the peers —
// in words 0 / 1 (ticks of 10 ns, waits), code 6 — the flag barriers — in words 2 / 3.  The first multi-GPU run reads them to tell exchange time from chain time.
__device__ __forceinline__ void shard_wait_flags(const unsigned *mine, const unsigned *perr, int n, unsigned epoch, int *err, int code) {
    const int t = threadIdx.x;
    const unsigned long long tw0 = wall_clock64();
    if (t < n) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        long spins = 0;
        // (relaxed polls: an acquire per poll is a buffer_inv per poll; what is read behind the wait is either read with system-scope atomics — the rows — or by the
        // next kernel, whose start is the acquire)
        while ((int)(__hip_atomic_load(mine + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if ((spins & 1023) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                         __hip_atomic_load(perr + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) { atomicCAS(err, 0, 9); break; }
            if (spins > (1L << 24)) { atomicCAS(err, 0, code); break; }
        }
    }
    if (t == 0) {                                           // (reconverged: the longest of the n waits)
        unsigned long long *st = reinterpret_cast<unsigned long long *>(err + 6) + (code == 7 ? 0 : 2);
        atomicAdd(st, wall_clock64() - tw0); atomicAdd(st + 1, 1ULL);
    }
}
// which: 0 = f1, 1 = f2, 2 = f3.  mode bit 0: signal every rank (this one included), bit 1: wait for every rank
__global__ __launch_bounds__(64) void shard_xbar_kernel(ShardPeers P, int which, int mode, unsigned epoch, int *err) {
    const int t = threadIdx.x;
    if (t < P.n && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)       // this rank has failed: tell every peer, so that none waits out its own timeout
        __hip_atomic_store(&P.x[t]->perr[P.me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((mode & 1) && t < P.n) {
        unsigned *f = which == 0 ? P.x[t]->f1 : which == 1 ? P.x[t]->f2 : P.x[t]->f3;
        // (relaxed: what the flag announces — the scatter of a round, the consumers' reads of a ring — was done by kernels that ENDED before this one started; a
        // release here is a write-back of the whole L2 per flag kernel: 5.4 us per round with one rank at 1 M)
        __hip_atomic_store(f + P.me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (mode & 2) {
        const unsigned *f = which == 0 ? P.x[P.me]->f1 : which == 1 ? P.x[P.me]->f2 : P.x[P.me]->f3;
        shard_wait_flags(f, P.x[P.me]->perr, P.n, epoch, err, 6);
    }
}

// SCAN of a shard, ONE launch (the two-level form of skel_k2_wide_kernel with the other ranks as a third level): workgroup j
// folds the rows of its TPW tiles per key (thread = key) and publishes the aggregate; the last workgroup to arrive folds the
// workgroups' aggregates into the RANK's row, stores it into every rank's exchange block and raises f1 there; every workgroup
// then waits until all ranks' rows have arrived here — prefix = fold of the rows of the ranks before this one and of this rank's
// workgroups before j — and writes the running prefix in front of each of its tiles, plus the totals over ALL ranks.
// All <= 64 workgroups of the launch are co-resident (they wait for the last of them).  Rows are indexed by global tile; this
// launch covers tiles w0 .. w0+Wl-1.
struct Sk2SArgs { const int2 *tbl; int2 *scan; int *total; int w0, Wl; unsigned long long *agg; unsigned *counter; unsigned target; unsigned epoch; int *err; };
template <int TPW, int CH = 8>
__global__ __launch_bounds__(SKK) void skel_k2s_kernel(Sk2SArgs g, ShardPeers P) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    __shared__ int s_last;
    const int t = threadIdx.x, j = blockIdx.x, r0 = j * TPW;
    int ac = 0, at = 0;
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (r0 + x0 + x < g.Wl) ? g.tbl[(size_t)(g.w0 + r0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) { at = v[x].x ? v[x].y : max(at, v[x].y); ac += v[x].x; }
    }
    __hip_atomic_store(g.agg + (size_t)j * SKK + t, ((unsigned long long)(unsigned)at << 32) | (unsigned)ac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_last = (__hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == g.target) ? 1 : 0;
    __syncthreads();
    const int nwg = (int)gridDim.x;
    if (s_last) {                                           // every workgroup's aggregate is out (agent scope): fold them into the rank's row
        int rc = 0, rt = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < nwg; i0 += 16) {
            unsigned long long pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pv[i] = (i0 + i < nwg) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32); rt = vc ? vt : max(rt, vt); rc += vc; }
        }
        const unsigned long long row = ((unsigned long long)(unsigned)rt << 32) | (unsigned)rc;
        for (int p = 0; p < P.n; ++p) __hip_atomic_store(&P.x[p]->ragg[P.me][t], row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);            // system scope: the row is out before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < P.n) __hip_atomic_store(&P.x[t]->f1[P.me], g.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // every rank's row of this round (this rank's own among them: its flag is raised by the last arriver above)
    shard_wait_flags(P.x[P.me]->f1, P.x[P.me]->perr, P.n, g.epoch, g.err, 7);
    __syncthreads();
    if (__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // incomplete rows: no output pass (no barrier follows)
    int ec = 0, et = 0, tot = 0;
    {
        unsigned long long rv[SHARD_MAX];
#pragma unroll
        for (int r = 0; r < SHARD_MAX; ++r) rv[r] = (r < P.n) ? __hip_atomic_load(&P.x[P.me]->ragg[r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ULL;
#pragma unroll
        for (int r = 0; r < SHARD_MAX; ++r) {
            const int vc = (int)(unsigned)rv[r], vt = (int)(rv[r] >> 32);
            tot += vc;
            if (r < P.me) { et = vc ? vt : max(et, vt); ec += vc; }
        }
    }
#pragma unroll 1
    for (int i0 = 0; i0 < j; i0 += 16) {
        unsigned long long pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = (i0 + i < j) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32); et = vc ? vt : max(et, vt); ec += vc; }
    }
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (r0 + x0 + x < g.Wl) ? g.tbl[(size_t)(g.w0 + r0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) {
            if (r0 + x0 + x < g.Wl) g.scan[(size_t)(g.w0 + r0 + x0 + x) * SKK + t] = make_int2(ec, ec ? et : -1);
            et = v[x].x ? v[x].y : max(et, v[x].y); ec += v[x].x;
        }
    }
    if (j == 0) g.total[t] = tot;
}

// The same scan WITHOUT waiting workgroups (round 5; the form skel_k2_local_kernel gave the plain engine in round 4): every workgroup writes the prefixes LOCAL to
// itself in its one pass over its rows, publishes its aggregate and arrives; only the LAST arriver goes on — it folds the workgroups' aggregates into the RANK's row,
// publishes that to every rank and raises f1, waits (bounded) for the other ranks' rows, and leaves one exclusive row per workgroup in aggx — the fold of the
// rows of the ranks before this one and of this rank's workgroups before it — plus the totals over ALL ranks.  The rank launch folds aggx[(tile - w0) / TPW] in
// front of its row (sk_fold_aggx).  One workgroup of the launch waits for the peers instead of all of them (22 us -> 8 us per round with one rank at 1 M).
template <int TPW, int CH = 16>
__global__ __launch_bounds__(SKK) void skel_k2s_local_kernel(Sk2SArgs g, ShardPeers P, int2 *aggx) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    __shared__ int s_last;
    const int t = threadIdx.x, j = blockIdx.x, r0 = j * TPW;
    int lc = 0, lt = 0;                                      // running local prefix for key t
#pragma unroll 1
    for (int x0 = 0; x0 < TPW; x0 += CH) {
        int2 v[CH];
#pragma unroll
        for (int x = 0; x < CH; ++x) v[x] = (r0 + x0 + x < g.Wl) ? g.tbl[(size_t)(g.w0 + r0 + x0 + x) * SKK + t] : make_int2(0, 0);
#pragma unroll
        for (int x = 0; x < CH; ++x) {
            if (r0 + x0 + x < g.Wl) g.scan[(size_t)(g.w0 + r0 + x0 + x) * SKK + t] = make_int2(lc, lt);
            lt = v[x].x ? v[x].y : max(lt, v[x].y); lc += v[x].x;
        }
    }
    __hip_atomic_store(g.agg + (size_t)j * SKK + t, ((unsigned long long)(unsigned)lt << 32) | (unsigned)lc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_last = (__hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == g.target) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const int nwg = (int)gridDim.x;                          // <= 64
    int rc = 0, rt = 0;                                      // this rank's row: the fold of its workgroups' aggregates
#pragma unroll 1
    for (int i0 = 0; i0 < nwg; i0 += 16) {
        unsigned long long pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = (i0 + i < nwg) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32); rt = vc ? vt : max(rt, vt); rc += vc; }
    }
    const unsigned long long row = ((unsigned long long)(unsigned)rt << 32) | (unsigned)rc;
    for (int p = 0; p < P.n; ++p) __hip_atomic_store(&P.x[p]->ragg[P.me][t], row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the row goes out as write-through system-scope stores; every storing wave drains them, then ONE relaxed flag store per peer (MI355X_MICROARCH.md, the
    // drained-flag form) — no release fence: a system-scope release writes the whole L2 back, and nothing but the row has to be visible with the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < P.n) __hip_atomic_store(&P.x[t]->f1[P.me], g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    shard_wait_flags(P.x[P.me]->f1, P.x[P.me]->perr, P.n, g.epoch, g.err, 7);      // every rank's row of this round (bounded; this rank's own among them)
    __syncthreads();
    if (__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // incomplete rows: no aggregates (the rank launch sees the error word and scatters nothing)
    int ec = 0, et = 0, tot = 0;
    {
        unsigned long long rv[SHARD_MAX];
#pragma unroll
        for (int r = 0; r < SHARD_MAX; ++r) rv[r] = (r < P.n) ? __hip_atomic_load(&P.x[P.me]->ragg[r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ULL;
#pragma unroll
        for (int r = 0; r < SHARD_MAX; ++r) {
            const int vc = (int)(unsigned)rv[r], vt = (int)(rv[r] >> 32);
            tot += vc;
            if (r < P.me) { et = vc ? vt : max(et, vt); ec += vc; }
        }
    }
#pragma unroll 1
    for (int i0 = 0; i0 < nwg; i0 += 16) {                   // (the aggregates again: L2)
        unsigned long long pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = (i0 + i < nwg) ? __hip_atomic_load(g.agg + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i0 + i < nwg) aggx[(size_t)(i0 + i) * SKK + t] = make_int2(ec, et);
            const int vc = (int)(unsigned)pv[i], vt = (int)(pv[i] >> 32);
            et = vc ? vt : max(et, vt); ec += vc;
        }
    }
    g.total[t] = tot;
}

// PULL: the consumer of rounds s0 .. s0+ns-1 of a batch copies those skeleton states (a, d, keys) out of every rank's skeleton
// ring — the range each rank owns — into slots slot_step * s of its own full ring.  grid (chunks, ns, ranks); 16 bytes per
// thread and array.  slot_step == 0 (with ns == 1): skeleton slot s0 into slot 0.
struct ShardPullArgs {
    const int *A[SHARD_MAX]; const int *D[SHARD_MAX]; const unsigned char *K[SHARD_MAX];   // slot 0 of the batch's SKELETON ring / key row 0 in every rank
    int *a; int *d; unsigned char *k;                                                      // slot 0 of the FULL ring / key row 0 in this rank
    size_t strideA, strideD, strideK;                                                      // per slot (ints) / per key row (bytes)
    int pb[SHARD_MAX + 1]; int n, me, M, s0, slot_step;
};
__global__ __launch_bounds__(BLOCK) void shard_pull_kernel(ShardPullArgs g) {
    const int o = blockIdx.z, s = g.s0 + blockIdx.y;
    const int lo = g.pb[o], hi = g.pb[o + 1];               // multiples of 256 except the last rank's end (= M)
    const size_t dst = (size_t)s * g.slot_step;
    const int4 *sa = reinterpret_cast<const int4 *>(g.A[o] + (size_t)s * g.strideA), *sd = reinterpret_cast<const int4 *>(g.D[o] + (size_t)s * g.strideD);
    int4 *da = reinterpret_cast<int4 *>(g.a + dst * g.strideA), *dd = reinterpret_cast<int4 *>(g.d + dst * g.strideD);
    const int hiD = (o == g.n - 1) ? hi + 1 : hi;           // d[M], the closing sentinel, lives with the last rank
    for (int i = lo / 4 + blockIdx.x * BLOCK + threadIdx.x; i < (hiD + 3) / 4; i += gridDim.x * BLOCK) {
        if (i < (hi + 3) / 4) da[i] = sa[i];
        dd[i] = sd[i];
    }
    if (g.slot_step == 0 && o != g.me) return;              // the keys of a pulled slot 0 are re-derived (pass start / replicated batch)
    const uint4 *sk = reinterpret_cast<const uint4 *>(g.K[o] + (size_t)s * g.strideK);
    uint4 *dk = reinterpret_cast<uint4 *>(g.k + (size_t)s * g.strideK);
    if (o != g.me) for (int i = lo / 16 + blockIdx.x * BLOCK + threadIdx.x; i < (hi + 15) / 16; i += gridDim.x * BLOCK) dk[i] = sk[i];
}

// RANK (K3): per tile — stable rank of every position among its key (ballot refinement inside
// 64-position chunks + a per-key scan over the chunks), previous same-key position, range max of d_k
// through a sparse table in LDS, scatter of (a | next allele tag, d', next key).
// TR > 0 (two-launch round, W <= TR tiles): the per-key scan over the tiles is done here, from the
// table: W coalesced 8-byte loads per thread, issued first and consumed last, behind the
// ballot refinement and the sparse table.  TR == 0: before/carry/total come from skel_k2_kernel.
constexpr int SKN_MAXW = 128;
// R4 (wide panels: more tiles than fit the chip at once): the range maxima come from a radix-4 sparse table (windows 1, 4, 16, 64,
// 256; <= 4 reads per query instead of 2) — 10 KB instead of 18 at T = 512, 22 KB per workgroup instead of 30: 7 workgroups per
// CU instead of 5, so the 1954 tiles of M = 1 M almost fit in one round (1792 resident) instead of needing two (1280).
template <int EPT, int TR, bool R4, bool SHARD, int MODE = SKM_LAUNCH>
__device__ __forceinline__ void skel_rank_body(const SkArgs &g, const SkShardOut *so, int wsel = -1) {
    static_assert(MODE == SKM_LAUNCH || (TR == 0 && !SHARD), "the persistent forms take the scanned tables");
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);                          // the dependent chain shares SIMDs with the throughput kernels of the consumer stream: issue first
#endif
    constexpr int T = BLOCK * EPT, NC = EPT * WAVES;        // positions per tile, 64-position chunks per tile
    constexpr int NL = R4 ? ((EPT == 1) ? 4 : 5) : ((EPT == 4) ? 10 : (EPT == 2) ? 9 : 8);   // sparse table levels: windows 1 .. T/2 (radix 2) or 1 .. 4^(NL-1) (radix 4)
    __shared__ short s_cnt[NC][SKK];                        // per chunk: count -> base (exclusive over chunks)
    __shared__ short s_lastp[NC][SKK];                      // per chunk: last local position of the key -> previous one before the chunk
    __shared__ int s_tbl[NL][T];                            // s_tbl[l][i] = max d over (i-2^l, i]
    // (round 4) 20 480 bytes at 512 positions, so that EIGHT workgroups fit a CU's 160 KB (22 560 held seven: at 1 M haplotypes 1 792 of the 1 954 tiles
    // were resident and the rest made a second, nearly empty round of the chip): per key ONE destination base (bucket base + keys before the tile) and ONE
    // first-occurrence word (carry | SK_EFLAG, or the final value); the cross-wave scan of the totals borrows eight words of the sparse table's top level
    // that no query reads (a window of 4^(NL-1) positions never ends below index 4^(NL-1) - 1; those eight are not written there either).
    __shared__ int s_base[SKK], s_ext[SKK];
    constexpr int SK_EFLAG = 0x40000000;
    int *const s_gw = R4 ? &s_tbl[NL - 1][0] : nullptr, *const s_lw = R4 ? &s_tbl[NL - 1][WAVES] : nullptr;
    __shared__ int s_gwx[R4 ? 1 : WAVES], s_lwx[R4 ? 1 : WAVES];
    __shared__ int *s_pa[SHARD ? SHARD_MAX : 1], *s_pd[SHARD ? SHARD_MAX : 1]; __shared__ unsigned char *s_pk[SHARD ? SHARD_MAX : 1];
    __shared__ int s_pb[SHARD ? SHARD_MAX : 1];
    __shared__ int s_failed;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id(), w = (wsel >= 0) ? wsel : g.w0 + ((g.xcd & 2) ? xcd_tile(blockIdx.x, g.W) : blockIdx.x);
    const int S = w * T;
    if constexpr (SHARD) { if (t == 0) s_failed = __hip_atomic_load(so->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // read by all after the first barrier
    if constexpr (SHARD) { if (t < SHARD_MAX) { s_pa[t] = so->a[t]; s_pd[t] = so->d[t]; s_pk[t] = so->k[t]; s_pb[t] = so->pb[t + 1]; } }   // visible after the barriers below
    int av[EPT], dv[EPT], key[EPT];
    unsigned nk[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {                         // striped: chunk r*4+wv = 64 consecutive positions
        const int i = S + r * BLOCK + t;
        av[r] = skm_ld<MODE>(g.a + i); dv[r] = skm_ld<MODE>(g.d + i); key[r] = (int)skm_ld<MODE>(g.keys + i);
    }
    int2 row[TR > 0 ? TR : 1];
    int bq = 0, cq = -1, tq = 0;
    if constexpr (TR > 0) {
#pragma unroll
        for (int r = 0; r < TR; ++r) row[r] = (r < g.W) ? g.tbl[(size_t)r * SKK + t] : make_int2(0, 0);
    } else {
        const int srow = g.pair ? (w >> 1) : w;
        int2 sv = skm_ld2<MODE>(g.scan + (size_t)srow * SKK + t);
        if (g.aggx) sv = sk_fold_aggx(skm_ld2<MODE>(g.aggx + (size_t)((srow - g.w0) / g.aggx_tpw) * SKK + t), sv);      // (both loads in flight together; w0 = 0 outside the sharded chain, which has no pair rows)
        if (g.pair && (w & 1)) {                            // second tile of its pair: fold the first one's row in
            const int2 r0 = skm_ld2<MODE>(g.tbl0 + (size_t)(w >> 1) * SKK + t);
            sv.y = r0.x ? r0.y : (sv.x ? max(sv.y, r0.y) : -1);
            sv.x += r0.x;
        }
        bq = sv.x; cq = sv.y; tq = skm_ld<MODE>(g.total + t);
    }
    for (int x = t; x < NC * SKK / 2; x += BLOCK) { reinterpret_cast<int *>(&s_cnt[0][0])[x] = 0; reinterpret_cast<int *>(&s_lastp[0][0])[x] = -1; }
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int l = r * BLOCK + t;
        const bool valid = S + l < g.M;
        av[r] &= AMASK; if (!valid) { dv[r] = 0; key[r] = -1; }
        s_tbl[0][l] = dv[r];
        nk[r] = (g.has_next && valid && !g.ycnext) ? (unsigned)g.kbnext[av[r]] : 0u;   // next round's key (bit 0 = the output state's tag)
    }
    int rk[EPT], pl[EPT];
    const unsigned long long lt = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
    lds_barrier();                                          // zeroed tables visible
    if constexpr (SHARD) { if (s_failed) return; }          // a bounded wait ran out earlier in this pass: nothing more goes into the peers' rings
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        unsigned long long same = __ballot(key[r] >= 0);
#pragma unroll
        for (int b = 0; b < SKB; ++b) { const unsigned long long bal = __ballot((key[r] >> b) & 1); same &= ((key[r] >> b) & 1) ? bal : ~bal; }
        const unsigned long long before = same & lt;
        rk[r] = __popcll(before);
        pl[r] = before ? (r * 4 + wv) * 64 + (63 - __clzll(before)) : -1;
        if (key[r] >= 0 && !before) {                       // leader of its key in this chunk
            s_cnt[r * 4 + wv][key[r]] = (short)__popcll(same);
            s_lastp[r * 4 + wv][key[r]] = (short)((r * 4 + wv) * 64 + (63 - __clzll(same)));
        }
    }
    lds_barrier();
    {   // thread q = key: exclusive scan over the chunks
        int base = 0, last = -1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cn = s_cnt[c][t], lp = s_lastp[c][t];
            s_cnt[c][t] = (short)base; s_lastp[c][t] = (short)last;
            base += cn; if (cn) last = lp;
        }
    }
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        lds_barrier();
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * BLOCK + t;
            if (R4) {
                const int wq = 1 << (2 * (l - 1));
                int m = s_tbl[l - 1][i];
                if (i - wq >= 0) m = max(m, s_tbl[l - 1][i - wq]);
                if (i - 2 * wq >= 0) m = max(m, s_tbl[l - 1][i - 2 * wq]);
                if (i - 3 * wq >= 0) m = max(m, s_tbl[l - 1][i - 3 * wq]);
                if (l < NL - 1 || i >= 2 * WAVES) s_tbl[l][i] = m;        // (the top level's first eight words carry the totals' scan, see above)
            } else {
                const int j = i - (1 << (l - 1));
                s_tbl[l][i] = (j >= 0) ? max(s_tbl[l - 1][i], s_tbl[l - 1][j]) : s_tbl[l - 1][i];
            }
        }
    }
    if constexpr (TR > 0) {   // thread q = key: scan of the tiles (keys before this tile, carry = max d since the key's last earlier occurrence, total)
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const int c = row[r].x, tl = row[r].y;
            if (r < w) { cq = c ? tl : (cq >= 0 ? max(cq, tl) : -1); bq += c; }
            tq += c;
        }
        g.scan[(size_t)w * SKK + t] = make_int2(bq, cq);   // kept for the fill kernel
        if (w == 0) g.total[t] = tq;
    }
    // bucket bases G (exclusive prefix of the key totals) and the nearest lower non-empty key
    const int ginc = wave_iscan_sum(tq), linc = wave_iscan_max(tq ? t + 1 : 0);
    int *const gwp = R4 ? s_gw : s_gwx, *const lwp = R4 ? s_lw : s_lwx;
    if (lane == 63) { gwp[wv] = ginc; lwp[wv] = linc; }
    const int lexc = lane_shr1(linc, 0);
    lds_barrier();
    int Gq = ginc - tq, lq = lexc;
    for (int x = 0; x < wv; ++x) { Gq += gwp[x]; lq = max(lq, lwp[x]); }
    lq -= 1;
    s_base[t] = Gq + bq;
    s_ext[t] = (cq >= 0) ? (cq | SK_EFLAG) : (lq >= 0 ? g.k + 1 + (31 - __clz(t ^ lq)) : 0);      // a first occurrence in the tile: the carry (max with the local maximum) or the key-difference value
    lds_barrier();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        if (key[r] < 0) continue;
        const int l = r * BLOCK + t, c = r * 4 + wv, ky = key[r];
        const int rank = s_cnt[c][ky] + rk[r];
        const int p = (pl[r] >= 0) ? pl[r] : s_lastp[c][ky];        // previous same-key position in the tile, or -1
        // range max of d over (p, l]  (p = -1: the whole prefix): two windows of 2^lv >= len/2
        const int len = l - p;
        int rm;
        if (R4) {
            const int lv = min((31 - __clz(len)) >> 1, NL - 1), wq = 1 << (2 * lv);
            rm = max(max(s_tbl[lv][l], s_tbl[lv][p + wq]), max(s_tbl[lv][len > 2 * wq ? l - wq : l], s_tbl[lv][len > 3 * wq ? l - 2 * wq : l]));
        } else {
            const int lv = min(31 - __clz(len), NL - 1);
            rm = max(s_tbl[lv][l], s_tbl[lv][p + (1 << lv)]);
        }
        int dd;
        if (p >= 0) dd = rm;
        else { const int ex = s_ext[ky]; dd = (ex & SK_EFLAG) ? max(ex & ~SK_EFLAG, rm) : ex; }
        const int pos = s_base[ky] + rank;
        if (pos == 0) dd = g.k + SKB + 1;                  // sentinel (pbwtCore.c:507 after the 8th site)
        if constexpr (SHARD) {                              // the owner of the destination: pb[o] <= pos < pb[o+1] (s_pb holds pb[1..]; unused entries INT_MAX)
            int o = 0;
#pragma unroll
            for (int x = 0; x < SHARD_MAX - 1; ++x) o += (pos >= s_pb[x]) ? 1 : 0;
            s_pa[o][pos] = av[r] | (int)((nk[r] & 1u) << 31);
            s_pd[o][pos] = dd;
            s_pk[o][pos] = (unsigned char)nk[r];
        } else if (g.ycnext) {                              // read side: the tag of a position is a bit of the sorted column, the keys were derived from the columns
            const unsigned tg = g.has_next ? (unsigned)((g.ycnext[pos >> 6] >> (pos & 63)) & 1ULL) : 0u;
            g.a_out[pos] = av[r] | (int)(tg << 31);
            g.d_out[pos] = dd;
        } else {
            skm_st<MODE>(g.a_out + pos, av[r] | (int)((nk[r] & 1u) << 31));
            skm_st<MODE>(g.d_out + pos, dd);
            skm_st<MODE>(g.keys_out + pos, (unsigned char)nk[r]);
        }
    }
    if (w == g.Wtot - 1 && t == 0) {
        if constexpr (SHARD) so->d[so->n - 1][g.M] = g.k + SKB + 1;      // d[M] lives with the last rank
        else skm_st<MODE>(g.d_out + g.M, g.k + SKB + 1);
    }
}
template <int EPT, int TR, bool R4 = false>
__global__ __launch_bounds__(BLOCK) void skel_rank_kernel(SkArgs g) { skel_rank_body<EPT, TR, R4, false>(g, nullptr); }
// position-sharded form: tiles w0 .. w0+W-1, scatter through the owners' table
template <int EPT, bool R4>
__global__ __launch_bounds__(BLOCK) void skel_rank_shard_kernel(SkArgs g, SkShardOut so) { skel_rank_body<EPT, 0, R4, true>(g, &so); }

// ---------------------------------------------------------------------------------------------
// THE ONE-LAUNCH ROUND (round 5).  The three-launch round pays three device-wide dependencies per 8 sites (hist -> scan -> rank: 3.6-3.9 us each at
// <= 100 k haplotypes, 0.5 us of data).  Two observations remove two of them:
//  (1) total[key] of a round — and with it every bucket base and the "nearest lower non-empty key" — does not depend on the ORDER: it is the number of
//      haplotypes whose alleles at the round's 8 sites spell that key, a histogram of one byte plane of the transposed panel.  skel_totals_kernel
//      computes it for every round of a batch at once, off the chain.
//  (2) what is left of the scan — per tile and key the keys before the tile and the carry — is a prefix over the tiles IN ORDER, and a tile can
//      fetch it from its predecessors inside the launch (decoupled look-back, two levels): a tile publishes its {count, tail} row as 256 tagged 8-byte
//      granules (ONE sc1 store each: the data is its own flag — MI355X_MICROARCH.md, the R2 form), folds the rows of the tiles before it in its group of
//      g1 tiles; the LAST tile of a group publishes the group's aggregate row; every tile folds the aggregates of the groups before its own.  Two
//      hand-offs of <= g1 - 1 and <= W / g1 - 1 rows (13 + 13 at 196 tiles) instead of two kernel boundaries and two passes over a 400 KB table.
// The kernel is skel_rank_body with the tile's own row derived from tables it builds anyway (per-key count and last position from the chunk scan, the
// tail as a range maximum from the sparse table) and the look-back in front of the bucket bases.  Waits are bounded by the wall clock (error 11); every
// workgroup of the launch must be able to become resident (W <= 1024 and the occupancy check in pbwtamd_engine_create): a tile waits for tiles
// before it, which the xcd_tile dealing does not dispatch in tile order.
// Granule: tail [0, 31) | count [32, 53) | tag [53, 64): the tail is the low word as it stands, count and tag share the high word.
__device__ __forceinline__ unsigned long long sk1_enc(int c, int tl, unsigned tag) { return (unsigned long long)(unsigned)tl | ((unsigned long long)((unsigned)c | ((tag & 2047u) << 21)) << 32); }
__device__ __forceinline__ int sk1_cnt(unsigned long long v) { return (int)((unsigned)(v >> 32) & 0x1fffffu); }
__device__ __forceinline__ int sk1_tail(unsigned long long v) { return (int)(unsigned)v; }
// bits that differ from the wanted tag, accumulated over granules: all tags right <=> the accumulated word is below 2^21
__device__ __forceinline__ unsigned sk1_tagdiff(unsigned long long v, unsigned want21) { return (unsigned)(v >> 32) ^ want21; }
__device__ __forceinline__ void sk1_fold1(unsigned long long v, int &c, int &tl) { const int vc = sk1_cnt(v), vt = sk1_tail(v); tl = vc ? vt : max(tl, vt); c += vc; }
// fold the n rows base[i * SKK + t], i = 0 .. n - 1 (in that order) onto (c, tl) with the scan's combine; false: the bounded wait ran out.  Loads, tag checks
// and folds run in blocks of four rows behind wave-uniform guards: nothing is spent on rows beyond n
template <int CH>
__device__ __forceinline__ bool sk1_fold_rows(const unsigned long long *base, int n, unsigned tag, int &c, int &tl, int *err, int code) {
    static_assert(CH % 4 == 0, "blocks of four rows");
    const int t = threadIdx.x;
    const unsigned want21 = (tag & 2047u) << 21;
#pragma unroll 1
    for (int i0 = 0; i0 < n; i0 += CH) {
        unsigned long long v[CH];
        int spins = 0; unsigned long long t0 = 0;
        for (;;) {
#pragma unroll
            for (int b = 0; b < CH; b += 4) {
                if (i0 + b < n) {
#pragma unroll
                    for (int i = b; i < b + 4; ++i) v[i] = (i0 + i < n) ? __hip_atomic_load(base + (size_t)(i0 + i) * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)want21 << 32);
                }
            }
            unsigned diff = 0;
#pragma unroll
            for (int b = 0; b < CH; b += 4) {
                if (i0 + b < n) {
#pragma unroll
                    for (int i = b; i < b + 4; ++i) diff |= sk1_tagdiff(v[i], want21);
                }
            }
            if (__all(diff < (1u << 21))) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0) {                      // bounded (2 s of wall clock, 100 MHz): a predecessor that never ran must not hang the GPU
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > 200000000ULL || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicCAS(err, 0, code); return false; }
            }
        }
#pragma unroll
        for (int b = 0; b < CH; b += 4) {
            if (i0 + b < n) {
#pragma unroll
                for (int i = b; i < b + 4; ++i) sk1_fold1(v[i], c, tl);     // beyond n: (0, 0), the identity
            }
        }
    }
    return true;
}
// both levels of a tile's look-back in ONE round trip (n1, n2 <= CH): the rows before the tile in its group onto (pc, pt), the aggregates of the groups before
// onto (qc, qt).  A level whose granules are not all there is polled again on its own (sk1_fold_rows).
template <int CH>
__device__ __forceinline__ bool sk1_fold_both(const unsigned long long *b1, int n1, const unsigned long long *b2, int n2, unsigned tag, int &pc, int &pt, int &qc, int &qt, int *err, int code) {
    static_assert(CH % 4 == 0, "blocks of four rows");
    const int t = threadIdx.x;
    const unsigned want21 = (tag & 2047u) << 21;
    const unsigned long long idv = (unsigned long long)want21 << 32;
    unsigned long long v1[CH], v2[CH];
#pragma unroll
    for (int b = 0; b < CH; b += 4) {
        if (b < n1) {
#pragma unroll
            for (int i = b; i < b + 4; ++i) v1[i] = (i < n1) ? __hip_atomic_load(b1 + (size_t)i * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : idv;
        }
    }
#pragma unroll
    for (int b = 0; b < CH; b += 4) {
        if (b < n2) {
#pragma unroll
            for (int i = b; i < b + 4; ++i) v2[i] = (i < n2) ? __hip_atomic_load(b2 + (size_t)i * SKK + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : idv;
        }
    }
    unsigned d1 = 0, d2 = 0;
#pragma unroll
    for (int b = 0; b < CH; b += 4) {
        if (b < n1) {
#pragma unroll
            for (int i = b; i < b + 4; ++i) d1 |= sk1_tagdiff(v1[i], want21);
        }
        if (b < n2) {
#pragma unroll
            for (int i = b; i < b + 4; ++i) d2 |= sk1_tagdiff(v2[i], want21);
        }
    }
    const bool a1 = __all(d1 < (1u << 21)), a2 = __all(d2 < (1u << 21));
    if (a1) {
#pragma unroll
        for (int b = 0; b < CH; b += 4) {
            if (b < n1) {
#pragma unroll
                for (int i = b; i < b + 4; ++i) sk1_fold1(v1[i], pc, pt);
            }
        }
    }
    if (a2) {
#pragma unroll
        for (int b = 0; b < CH; b += 4) {
            if (b < n2) {
#pragma unroll
                for (int i = b; i < b + 4; ++i) sk1_fold1(v2[i], qc, qt);
            }
        }
    }
    bool ok = true;
    if (!a1) ok = sk1_fold_rows<CH>(b1, n1, tag, pc, pt, err, code);
    if (!a2) ok = ok && sk1_fold_rows<CH>(b2, n2, tag, qc, qt, err, code);
    return ok;
}

// THE SCANNER FORM (round 6; wide panels).  A tile's look-back reads O(sqrt(W)) rows of 2 KB: at 1 954 tiles that is 344 MB of polls per round against 129 MB of
// algorithmic traffic, and the tiles cannot all be resident (which the look-back's XCD-contiguous dealing needs).  Here the per-key scan over the tiles is done
// INSIDE the launch by workgroups dispatched in FRONT of the tiles, laid out the way the chip is: a GROUP of g1 <= 32 consecutive tiles and its SCANNER workgroup
// sit on ONE XCD (workgroups go to the XCDs round robin: inside every run of 8 * g1 tile workgroups XCD x takes one group), and everything a tile touches of the
// scan stays inside that XCD's L2 — plain stores, L1-bypassing (nt) loads, the hand-off of skel_team_kernel (0.4 us).  Only 2 KB per group cross the XCDs, twice:
//   tile    : row[tile] (plain)                                       -> scanner of its group
//   scanner : pass 1 over the group's rows: the group's aggregate     -> aggregators           (sc1, laid out per aggregator workgroup)
//   aggregators (SK1_NAGG workgroups, lanes = groups): exclusive fold  -> scanner               (sc1; one poller per line)
//   scanner : pass 2 over the rows (its L2): the FINAL {keys before the tile, carry} of every tile, scanl[tile] (plain) -> tile: ONE granule per key, one hop
// What was measured on the way (profiles/r06_onepass.txt): 2 000 tiles polling cross-XCD granules with every lane are the fabric's whole request rate (every hand-off
// 3-5x slower); a line polled from several XCDs, or by 128 waves, before it is written takes the write 5-15 us instead of ~1; a branch per row in the scanner drains
// the load queue at every row (32 dependent round trips); a register spilled to scratch is a round trip.  Tile = dispatch order inside its XCD: a tile waits for its
// scanner, which waits for rows of its own group (the same run of workgroups) and for the aggregates of groups before it — deadlock-free whatever is resident
// (the device must hold the front and two runs of tiles: checked at pbwtamd_engine_create).
constexpr int SK1_NAGG = 8;                                 // aggregator workgroups in front of the scanners: SKK / (8 * WAVES) = 8 keys per wave
__host__ __device__ constexpr int sk1_front(int nscan) { return SK1_NAGG + (nscan + 7) / 8 * 8; }      // workgroups in front of the tiles (aggregators, scanners, padding to a multiple of 8: XCD of a tile = its index % 8)
__host__ __device__ constexpr int sk1_grid(int nscan, int W, int g1) { return sk1_front(nscan) + (W + 8 * g1 - 1) / (8 * g1) * (8 * g1); }
// wait until every lane's granule *p carries the tag; LOCAL: the writer is on this XCD (plain store, nt load), otherwise an agent-scope (sc1) load
template <bool LOCAL>
__device__ __forceinline__ bool sk1_wait_row(const unsigned long long *p, unsigned want21, int *err, int code, unsigned long long *val = nullptr) {
    int spins = 0; unsigned long long t0 = 0;
    for (;;) {
        const unsigned long long v = LOCAL ? __builtin_nontemporal_load(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(sk1_tagdiff(v, want21) < (1u << 21))) { if (val) *val = v; return true; }
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > 200000000ULL || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicCAS(err, 0, code); return false; }
        }
    }
}
// A SCANNER (thread = key).  Rows stream through a window of WIN buffer loads in flight: a load costs its two data registers and nothing for its address, rows
// beyond the group are out of range (zeros), and the pass is STRAIGHT-LINE code, so that the compiler counts the loads in flight exactly.  Pass 1 folds; a granule
// that had not landed when it was read stops the fold, and the pass restarts there behind a watch on exactly that row (each wave for its own 64 keys).
template <int WIN, int GMAX>
__device__ __forceinline__ bool sk1_scanner(const SkArgs &g, int sg) {
    const int t = threadIdx.x;
    const int f0 = sg * g.g1, n = min(min(g.g1, GMAX), g.W - f0);
    const unsigned want21 = (g.tag & 2047u) << 21;
    const unsigned long long idv = (unsigned long long)want21 << 32;
    const unsigned long long *rows = g.rows + (size_t)f0 * SKK;
    unsigned long long *out = g.scanl + (size_t)f0 * SKK;
    // (loads: the row's offset rides in the scalar offset and nothing is out of range — rows beyond the group are other groups' rows, or the 64 rows of slack behind
    // the table; pass 1 ends at the group's last row by count, pass 2's stores beyond it are out of range)
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long *>(rows), 0, (GMAX + WIN + 1) * SKK * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * SKK * 8, 0x00020000);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define SKS_STAMP(i) do { if (g.prof && t == 0 && sg < 64) g.prof[(size_t)(g.W + sg) * 8 + (i)] = wall_clock64(); } while (0)
    int c = 0, tl = 0, done = 0, stuck = 0;
#pragma unroll 1
    for (int watch = n - 1; done < n; watch = done) {
        if (!sk1_wait_row<true>(rows + (size_t)watch * SKK + t, want21, g.err, 11)) return false;
        if (done == 0) SKS_STAMP(2);
        asm volatile("" ::: "memory");
        const int voff = t * 8, soff = done * SKK * 8, left = n - done;
        u32x2 v[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rl, voff, soff + i * SKK * 8, 2 /* nt: this XCD's L2 */);
        int good = 1, adv = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < GMAX; i0 += WIN) {            // (a real loop: unrolled over all GMAX rows the compiler hoisted every load to the top and spilled them)
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                const unsigned long long x = ((unsigned long long)v[j].y << 32) | v[j].x;
                v[j] = __builtin_amdgcn_raw_buffer_load_b64(rl, voff, soff + (i0 + j + WIN) * SKK * 8, 2);
                good &= (int)(__ballot(sk1_tagdiff(x, want21) >= (1u << 21)) == 0ULL) & (int)(i0 + j < left);      // (no &&: a branch per row makes the compiler drain the load queue at every row)
                int nc = c, nt = tl;
                sk1_fold1(x, nc, nt);
                c = good ? nc : c; tl = good ? nt : tl; adv += good;
                __builtin_amdgcn_sched_barrier(0);          // (row by row: the scheduler otherwise pulls every row's unpacking to the top, waits for the whole window at once and spills)
            }
        }
        if (g.prof && t == 0 && sg < 64 && adv < left) g.prof[(size_t)(g.W + sg) * 8 + 3] += 1;
        if (adv == 0 && ++stuck > 64) { atomicCAS(g.err, 0, 11); return false; }      // (the watched row was there and the pass did not take it: cannot happen; never spin on it)
        done += adv;
    }
    // the group's aggregate, laid out per AGGREGATOR workgroup (a = key / 32): [a][group][32 keys] — every line of it is polled by one workgroup, on one XCD
    __hip_atomic_store(g.grows + ((size_t)(t >> 5) * g.nscan + sg) * 32 + (t & 31), sk1_enc(c, tl, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SKS_STAMP(1);
    // the fold of the groups before this one: one poller per line
    unsigned long long ex = idv;
    if (sg > 0 && !sk1_wait_row<false>(g.grows + (size_t)(g.nscan + sg) * SKK + t, want21, g.err, 11, &ex)) return false;
    SKS_STAMP(6);
    asm volatile("" ::: "memory");
    // pass 2: the rows again (they are all there), the running GLOBAL prefix in front of each — scanl[tile][key] = {keys before the tile, max d since the key's last
    // occurrence before the tile (all tiles' maximum so far if there is none)}: the tile reads (count, count ? tail : -1)
    c = sk1_cnt(ex); tl = sk1_tail(ex);
    {
        const int voff = t * 8;
        u32x2 v[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rl, voff, i * SKK * 8, 2);
#pragma unroll 1
        for (int i0 = 0; i0 < GMAX; i0 += WIN) {
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                const unsigned long long x = ((unsigned long long)v[j].y << 32) | v[j].x;      // (beyond n: zeros, the identity; the store is out of range too)
                v[j] = __builtin_amdgcn_raw_buffer_load_b64(rl, voff, (i0 + j + WIN) * SKK * 8, 2);
                const unsigned long long e = sk1_enc(c, tl, g.tag);
                u32x2 ev; ev.x = (unsigned)e; ev.y = (unsigned)(e >> 32);
                __builtin_amdgcn_raw_buffer_store_b64(ev, rs, voff + (i0 + j) * SKK * 8, 0, 0);
                sk1_fold1(x, c, tl);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    SKS_STAMP(4);
#undef SKS_STAMP
    return true;
}
// AN AGGREGATOR: the groups' aggregates grows[0 .. nscan) -> their exclusive folds grows[nscan + s].  A wave owns KPV keys, its LANES are the scanners (TPL <= 2
// consecutive ones per lane: up to 128 groups): one load per (key, scanner) in flight at once, a DPP scan across the lanes with the scan's (non-commutative)
// combine on the pair, one round trip behind the last aggregate — where a scanner folding the aggregates before its own (thread = key) walks up to 61 rows in batches.
// PROGRESSIVE: the exclusive row of group s goes out as soon as the aggregates of the groups before it are there — a tile waits for nothing dispatched after
// its own group (dispatch order, deadlock-free whatever is resident)
__device__ __forceinline__ void sk1_pair_scan(int &c, int &tl) {      // inclusive scan over the 64 lanes of (count, tail) under fold(L, R) = (L.c + R.c, R.c ? R.tl : max(L.tl, R.tl)); (0, 0) is the identity
#define SK1_STAGE(CTRL, MASK) { const int c2 = dpp_mov<CTRL, MASK>(0, c), t2 = dpp_mov<CTRL, MASK>(0, tl); tl = c ? tl : max(t2, tl); c += c2; }
    SK1_STAGE(0x111, 0xf) SK1_STAGE(0x112, 0xf) SK1_STAGE(0x114, 0xf) SK1_STAGE(0x118, 0xf) SK1_STAGE(0x142, 0xa) SK1_STAGE(0x143, 0xc)
#undef SK1_STAGE
}
template <int KPV>
__device__ __forceinline__ bool sk1_aggregator(const SkArgs &g, int gw) {
    static_assert(KPV % 2 == 0, "16-byte pieces of two keys");
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int lane = lane_id(), k0 = gw * KPV, ag = gw / (32 / KPV), kk0 = (gw % (32 / KPV)) * KPV;      // aggregator workgroup, first key inside its 32
    const unsigned want21 = (g.tag & 2047u) << 21;
    const int tpl = (g.nscan + 63) >> 6, s0 = lane * tpl;
    unsigned long long *dst = g.grows + (size_t)(g.nscan + s0) * SKK + k0;
    const int nl = (g.nscan + tpl - 1) / tpl;               // lanes that hold scanners
    // the aggregates through a descriptor over grows[0 .. nscan) ([aggregator][group][32 keys]): 16-byte loads (two keys); scanners beyond nscan are read OUT OF RANGE —
    // zeros, the identity of the fold — so that nothing has to be selected afterwards (registers: the kernel is held to 64, a spill to scratch costs a round trip)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.grows, 0, g.nscan * SKK * 8, 0x00020000);
    constexpr int OOB = 0x40000000;
    const bool in0 = s0 < g.nscan, in1 = tpl > 1 && s0 + 1 < g.nscan;
    const int vo0 = in0 ? ((ag * g.nscan + s0) * 32 + kk0) * 8 : OOB, vo1 = in1 ? ((ag * g.nscan + s0 + 1) * 32 + kk0) * 8 : OOB;
    int pub = -1;                                           // lanes 0 .. pub have published their exclusive rows
    int spins = 0; unsigned long long t0 = 0;
    while (pub < nl - 1) {
        asm volatile("" ::: "memory");
        // the watch: the first two keys of every scanner of this lane; the rest follow when those say that more groups are complete than have been published
        u32x4 q0[KPV / 2], q1[KPV / 2];
        q0[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo0, 0, 16 /* sc1 */);
        q1[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo1, 0, 16);
        const bool g0 = !in0 || ((q0[0].y ^ want21) | (q0[0].w ^ want21)) < (1u << 21), g1 = !in1 || ((q1[0].y ^ want21) | (q1[0].w ^ want21)) < (1u << 21);
        const unsigned long long gbad = ~__ballot(g0 && g1);
        const int glc = gbad ? (__ffsll((long long)gbad) - 1) : 64;
        if (min(glc, nl - 1) > pub) {
#pragma unroll
            for (int h = 1; h < KPV / 2; ++h) { q0[h] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + h * 16, 0, 16); q1[h] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo1 + h * 16, 0, 16); }
            unsigned d0 = 0, d1 = 0;
#pragma unroll
            for (int h = 0; h < KPV / 2; ++h) { d0 |= (q0[h].y ^ want21) | (q0[h].w ^ want21); d1 |= (q1[h].y ^ want21) | (q1[h].w ^ want21); }
            const bool ok0 = !in0 || d0 < (1u << 21), ok1 = !in1 || d1 < (1u << 21);
            const unsigned long long bad = ~__ballot(ok0 && ok1);
            const int lc = bad ? (__ffsll((long long)bad) - 1) : 64;       // lanes 0 .. lc-1 hold complete aggregates: lanes <= lc have everything before them
            const int upto = min(lc, nl - 1);
            if (upto > pub) {
                const bool mine = lane > pub && lane <= upto && in0, mine1 = mine && in1 && (lane < lc || ok0);
#pragma unroll
                for (int k = 0; k < KPV; ++k) {
                    const unsigned a_lo = (k & 1) ? q0[k / 2].z : q0[k / 2].x, a_hi = (k & 1) ? q0[k / 2].w : q0[k / 2].y;
                    const unsigned b_lo = (k & 1) ? q1[k / 2].z : q1[k / 2].x, b_hi = (k & 1) ? q1[k / 2].w : q1[k / 2].y;
                    const int ac = (int)(a_hi & 0x1fffffu), at = (int)a_lo, bc = (int)(b_hi & 0x1fffffu), bt = (int)b_lo;
                    // this lane's scanners, in order (incomplete lanes: the identity — nothing behind them is published)
                    int c = (lane < lc) ? ac + bc : 0, tl = (lane < lc) ? (bc ? bt : max(at, bt)) : 0;
                    sk1_pair_scan(c, tl);
                    int ec = lane_shr1(c, 0), et = lane_shr1(tl, 0);
                    if (mine) __hip_atomic_store(dst + k, sk1_enc(ec, et, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (mine1) { et = ac ? at : max(et, at); ec += ac; __hip_atomic_store(dst + SKK + k, sk1_enc(ec, et, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                }
                // (tpl == 2: lane lc's second scanner needs the lane's own first aggregate; if that was missing it goes out with the next pass — the lane is not counted as published)
                pub = (tpl > 1 && upto == lc && lc < 64 && !__builtin_amdgcn_readlane((int)ok0, min(lc, 63))) ? upto - 1 : upto;
                if (pub >= nl - 1) break;
                continue;                                   // more may have landed meanwhile: look again at once
            }
        }
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > 200000000ULL || __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicCAS(g.err, 0, 11); return false; }
        }
    }
    if (g.prof && lane == 0 && gw < 32) g.prof[(size_t)(g.W + gw) * 8 + 7] = wall_clock64();
    return true;
}
// totals of every round of a batch: tot[r * strideT + key] = number of entries of src[r * strideSrc + 0 .. M) equal to key.  src = the byte planes of the
// transposed panel (keys by haplotype, build side) or the rounds' key rows (by position, read side).  grid (chunks of SKTOT_CHUNK entries, rounds); more than one
// chunk per round: atomics onto totals zeroed by skel_totals_zero_kernel.
__global__ __launch_bounds__(BLOCK) void skel_totals_zero_kernel(int *tot, size_t strideT) { tot[(size_t)blockIdx.x * strideT + threadIdx.x] = 0; }
constexpr int SKTOT_CHUNK = 16384;                          // entries per workgroup: 64 bytes per thread, all four 16-byte loads in flight
__global__ __launch_bounds__(BLOCK) void skel_totals_kernel(const unsigned char *src, size_t strideSrc, int M, int *tot, size_t strideT) {
    __shared__ int h[WAVES][SKK];                           // a private histogram per wave: LDS atomics of different waves do not collide
    const int t = threadIdx.x, wv = wave_id(), r = blockIdx.y;
    const unsigned char *p = src + (size_t)r * strideSrc;
    for (int x = t; x < WAVES * SKK; x += BLOCK) (&h[0][0])[x] = 0;
    const int lo = blockIdx.x * SKTOT_CHUNK, hi = min(M, lo + SKTOT_CHUNK);
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i0 = lo + (q * BLOCK + t) * 16; v[q] = (i0 < hi) ? *reinterpret_cast<const uint4 *>(p + i0) : make_uint4(0, 0, 0, 0); }   // rows are padded to 4096 entries: whole 16-byte pieces
    lds_barrier();
    // the all-zero key (most positions of a real panel: a site's minor allele is rare) is counted in a register — same-address LDS atomics serialise; only the
    // other keys go through the LDS histogram
    int zeros = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i0 = lo + (q * BLOCK + t) * 16;
        const unsigned wd[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(max(hi - (i0 + 4 * j), 0), 4);  // entries of this word inside the panel
            if (wd[j] == 0u) { zeros += n; continue; }
#pragma unroll
            for (int b = 0; b < 4; ++b) { if (b >= n) break; const unsigned ky = (wd[j] >> (8 * b)) & 0xffu; if (ky) atomicAdd(&h[wv][ky], 1); else ++zeros; }
        }
    }
    const int wz = wave_sum(zeros);
    if (lane_id() == 0 && wz) atomicAdd(&h[wv][0], wz);
    lds_barrier();
    const int c = h[0][t] + h[1][t] + h[2][t] + h[3][t];
    if (gridDim.x == 1) tot[(size_t)r * strideT + t] = c;
    else if (c) atomicAdd(tot + (size_t)r * strideT + t, c);
}

template <int EPT, bool BOTH = false, bool MERGED = (EPT == 1), bool SCAN = false>
__device__ __forceinline__ void skel_onepass_body(const SkArgs &g) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int T = BLOCK * EPT, NC = EPT * WAVES;
    constexpr int NL = (EPT == 1) ? 4 : 5;                  // radix-4 sparse table: windows 1 .. 4^(NL-1) (EPT <= 2)
    static_assert(EPT <= 2, "the one-launch round runs 256- and 512-position tiles");
    __shared__ short s_cnt[NC][SKK];
    __shared__ short s_lastp[NC][SKK];
    __shared__ int s_tbl[NL][T];
    // SCAN (wide panels): 20 480 bytes exactly, so that EIGHT workgroups fit a CU's 160 KB (skel_rank_body's rule) — the failure word and the chunk maxima live in
    // words nothing else holds at the time: the top level of the sparse table is read from index 4^(NL-1) - 1 on only (its first 16 words are not written
    // there), and s_base is written behind the look-back
    static_assert(!SCAN || (MERGED && !BOTH), "the scanner form takes its row from the chunk tables");
    __shared__ int s_base[SKK + (SCAN ? 0 : 1 + NC)], s_ext[SKK];
    int &s_failed = SCAN ? s_tbl[NL - 1][2 * WAVES] : s_base[SKK];
    int *const s_cmax = SCAN ? &s_base[0] : &s_base[SKK + 1];
    constexpr int TOPFREE = SCAN ? 16 : 2 * WAVES;
    constexpr int SK_EFLAG = 0x40000000;
    int *const s_gw = &s_tbl[NL - 1][0], *const s_lw = &s_tbl[NL - 1][WAVES];   // (eight words of the top level no query reads: skel_rank_body)
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    if constexpr (SCAN) {
        const int bi = (int)blockIdx.x;
        if (bi < SK1_NAGG) { sk1_aggregator<SKK / (SK1_NAGG * WAVES)>(g, bi * WAVES + wv); return; }
        if (bi < sk1_front(g.nscan)) {
            const int sg = bi - SK1_NAGG;
            if (sg >= g.nscan) return;                      // (padding: the tiles start at a multiple of 8)
            if (g.prof && t == 0 && sg < 64) g.prof[(size_t)(g.W + sg) * 8] = wall_clock64();
            sk1_scanner<16, 32>(g, sg);
            return;
        }
    }
    if (!SCAN && g.nfold && (int)blockIdx.x >= g.W) {
        // a FOLDER: the rows of one group -> the group's aggregate, and nothing else.  With the group's last tile doing this in front of its own tables that tile
        // was the launch's last to scatter (profiles/r05_onepass.txt, r5t: it waits here for the rows of an XCD that entered late, 8.2 against 7.45 us).
        // Measured on top, not kept (r5g): the folder also folding the aggregates before its group into a BASE row, so that a tile's look-back is one fold — the
        // base arrives a hop later than the tiles can fold the aggregates themselves: 1.20-1.23 against 1.155 us/site at 100 k.
        // (nfold may be a multiple of the number of groups: the copies of a group's folder start their polls a third of a round trip apart and publish the same words)
        const int ngrp = (g.W + g.g1 - 1) / g.g1, fq = (int)blockIdx.x - g.W, fg = fq % ngrp, f0 = fg * g.g1;
        for (int z = fq / ngrp; z > 0; --z) __builtin_amdgcn_s_sleep(12);
        int fc = 0, ft = 0;
        if (sk1_fold_rows<16>(g.rows + (size_t)f0 * SKK, min(g.g1, g.W - f0), g.tag, fc, ft, g.err, 11))
            __hip_atomic_store(g.grows + (size_t)fg * SKK + t, sk1_enc(fc, ft, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // scanner form: workgroups go to the XCDs round robin; inside every run of 8 * g1 of them XCD x takes the g1 tiles of ONE group (and that group's scanner sits on
    // the same XCD): a group's exclusive aggregate is polled from one XCD only, and neighbouring tiles complete their destination lines in one L2 (xcd_tile's point)
    int wsc = 0;
    if constexpr (SCAN) {
        const int i = (int)blockIdx.x - sk1_front(g.nscan), run = 8 * g.g1, r = i % run;
        wsc = (i / run) * run + (r & 7) * g.g1 + (r >> 3);
        if (wsc >= g.W) return;                             // (the launch's last run is padded)
    }
    const int w = SCAN ? wsc : (g.xcd & 2) ? xcd_tile(blockIdx.x, g.W) : blockIdx.x;
    const int S = w * T;
#define SK1_STAMP(i) do { if (g.prof && t == 0) g.prof[(size_t)w * 8 + (i)] = wall_clock64(); } while (0)
    // MERGED (256-position tiles): the tile's row out of the rank's own chunk tables; otherwise the separate histogram phase (skel_hist_row) in front of them.
    // Measured (profiles/r05_onepass.txt, r5v2 / r5v3): merged 0.93 against 0.975 us/site at 50 k (1.01 / 1.046 beside the consumers); on 512-position tiles the
    // tiles then reach their look-back 0.7 us EARLIER than the folders' aggregates and pay a failed poll: 1.10-1.13 against 1.046 at 100 k (1.24-1.27 / 1.20)
    const int tp = MERGED ? ((t & ~63) | (63 - lane)) : t;  // the position (inside the tile's 256-position row r) this thread holds
    int av[EPT], dv[EPT], key[EPT], rk[EPT], pl[EPT];
    unsigned nk[EPT];
    int tq = 0, cnt_t = 0, tail_t = 0;
    const int grp = w / g.g1, first = grp * g.g1, lastw = min(first + g.g1, g.W) - 1;
    int pc = 0, pt = 0;
    bool ok = true;
    const bool selffold = !SCAN && (w == lastw) && g.nfold == 0;
    if constexpr (MERGED) {
        SK1_STAMP(0);
        // every address-known load of the launch is issued here.  A wave holds its 64-position chunks in REVERSE lane order (lane i = position 63 - i of the chunk):
        // a prefix scan over the lanes is then a suffix scan over the positions, which is what the row's "max of d after the key's last occurrence" needs — the
        // row comes out of the SAME ballots and the SAME per-key scan over the chunks that the rank's tables are built from (one LDS barrier behind the loads),
        // not from a separate histogram phase with its own loads, ballots and atomics (skel_hist_row: three barriers; -0.6 us of a tile's 5 us)
#pragma unroll
        for (int r = 0; r < EPT; ++r) { const int i = S + r * BLOCK + tp; av[r] = g.a[i]; dv[r] = g.d[i]; key[r] = (int)g.keys[i]; }
        tq = g.total[t];                                        // precomputed (skel_totals_kernel)
        if (!SCAN && t == 0) s_failed = 0;
        for (int x = t; x < NC * SKK / 2; x += BLOCK) { reinterpret_cast<int *>(&s_cnt[0][0])[x] = 0; reinterpret_cast<int *>(&s_lastp[0][0])[x] = -1; }
        int *const s_sfx = &s_tbl[NL - 1][0];                   // [T] max of d over the LATER positions of the element's chunk (the top level is written three barriers on)
        lds_barrier();                                          // zeroed tables visible (the loads are still in flight)
        // (1) chunk tables: every element's rank among the same keys of its chunk, its previous same-key position; per (chunk, key) count and last position
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int l = r * BLOCK + tp;
            const bool valid = S + l < g.M;
            av[r] &= AMASK; if (!valid) { dv[r] = 0; key[r] = -1; }
            s_tbl[0][l] = dv[r];
            nk[r] = (g.has_next && valid && !g.ycnext) ? (unsigned)g.kbnext[av[r]] : 0u;
        }
        const unsigned long long gt = (lane == 63) ? 0ULL : (~0ULL << (lane + 1));     // the lanes above mine = the positions before mine
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int cb = (r * 4 + wv) * 64;
            unsigned long long same = __ballot(key[r] >= 0);
#pragma unroll
            for (int b = 0; b < SKB; ++b) { const unsigned long long bal = __ballot((key[r] >> b) & 1); same &= ((key[r] >> b) & 1) ? bal : ~bal; }
            const unsigned long long before = same & gt;
            rk[r] = __popcll(before);
            pl[r] = before ? cb + 63 - (__ffsll((long long)before) - 1) : -1;
            if (key[r] >= 0 && !before) {                       // first of its key in this chunk
                s_cnt[r * 4 + wv][key[r]] = (short)__popcll(same);
                s_lastp[r * 4 + wv][key[r]] = (short)(cb + 63 - (__ffsll((long long)same) - 1));
            }
            const int inc = wave_iscan_max(dv[r]);             // lanes 0 .. mine = my position and the later ones of the chunk
            s_sfx[r * BLOCK + tp] = lane_shr1(inc, 0);
            if (lane == 63) s_cmax[r * 4 + wv] = inc;
        }
        lds_barrier();
        // (2) thread = key: exclusive scan over the chunks (count before the chunk, last position before it) — and with it this tile's ROW: the key's count and
        // the max of d behind its last occurrence (the whole tile's for an absent key), out at once: everything a tile after this one waits for
        {
            int base = 0, last = -1, lastc = -1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cn = s_cnt[c][t], lp = s_lastp[c][t];
                s_cnt[c][t] = (short)base; s_lastp[c][t] = (short)last;
                base += cn; if (cn) { last = lp; lastc = c; }
            }
            int later = (last >= 0) ? s_sfx[last] : 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) if (c > lastc) later = max(later, s_cmax[c]);
            cnt_t = base; tail_t = later;
        }
        SK1_STAMP(1);
        if constexpr (SCAN) g.rows[(size_t)w * SKK + t] = sk1_enc(cnt_t, tail_t, g.tag);      // (its scanner is on this XCD: a plain store, read there past the L1)
        else __hip_atomic_store(g.rows + (size_t)w * SKK + t, sk1_enc(cnt_t, tail_t, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // without folders the launch's critical path runs through the LAST tile of every group: rows of its group -> the group's aggregate -> every later tile.
        // That tile folds its group and publishes the aggregate before anything else
        if (selffold) {
            ok = sk1_fold_rows<16>(g.rows + (size_t)first * SKK, w - first, g.tag, pc, pt, g.err, 11);
            if (ok) {
                const int ac = pc + cnt_t, at = cnt_t ? tail_t : max(pt, tail_t);
                __hip_atomic_store(g.grows + (size_t)grp * SKK + t, sk1_enc(ac, at, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        SK1_STAMP(0);
        // every address-known load of the launch is issued here: the tile in the rank's (striped) order, then — inside skel_hist_row — keys and d once more in
        // the hist's (reverse-blocked) order: the same lines, 5 bytes per position more through the L1
#pragma unroll
        for (int r = 0; r < EPT; ++r) { const int i = S + r * BLOCK + t; av[r] = g.a[i]; dv[r] = g.d[i]; key[r] = (int)g.keys[i]; }
        tq = g.total[t];                                        // precomputed (skel_totals_kernel)
        if (t == 0) s_failed = 0;
        for (int x = t; x < NC * SKK / 2; x += BLOCK) { reinterpret_cast<int *>(&s_cnt[0][0])[x] = 0; reinterpret_cast<int *>(&s_lastp[0][0])[x] = -1; }
        // (1) this tile's row first, the way skel_hist_kernel derives it (three LDS barriers), and out with it: everything a tile after this one waits for
        int c0u = 0, t0u = 0;
        skel_hist_row<EPT, false>(g, w, cnt_t, tail_t, c0u, t0u);
        SK1_STAMP(1);
        __hip_atomic_store(g.rows + (size_t)w * SKK + t, sk1_enc(cnt_t, tail_t, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the launch's critical path runs through the LAST tile of every group: rows of its group -> the group's aggregate -> every later tile.  That tile folds its
        // group and publishes the aggregate before anything else; the others build their tables first and find the rows waiting
        if (selffold) {
            ok = sk1_fold_rows<16>(g.rows + (size_t)first * SKK, w - first, g.tag, pc, pt, g.err, 11);
            if (ok) {
                const int ac = pc + cnt_t, at = cnt_t ? tail_t : max(pt, tail_t);
                __hip_atomic_store(g.grows + (size_t)grp * SKK + t, sk1_enc(ac, at, g.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (2) in the shadow of the hand-offs: the rank's tables — chunk tables, the sparse table of d, every element's rank inside the tile, its previous
        // same-key position and the range maximum since then; none of it depends on another tile
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int l = r * BLOCK + t;
            const bool valid = S + l < g.M;
            av[r] &= AMASK; if (!valid) { dv[r] = 0; key[r] = -1; }
            s_tbl[0][l] = dv[r];
            nk[r] = (g.has_next && valid && !g.ycnext) ? (unsigned)g.kbnext[av[r]] : 0u;
        }
        const unsigned long long lt = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
        lds_barrier();
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            unsigned long long same = __ballot(key[r] >= 0);
#pragma unroll
            for (int b = 0; b < SKB; ++b) { const unsigned long long bal = __ballot((key[r] >> b) & 1); same &= ((key[r] >> b) & 1) ? bal : ~bal; }
            const unsigned long long before = same & lt;
            rk[r] = __popcll(before);
            pl[r] = before ? (r * 4 + wv) * 64 + (63 - __clzll(before)) : -1;
            if (key[r] >= 0 && !before) {
                s_cnt[r * 4 + wv][key[r]] = (short)__popcll(same);
                s_lastp[r * 4 + wv][key[r]] = (short)((r * 4 + wv) * 64 + (63 - __clzll(same)));
            }
        }
        lds_barrier();
        {   // thread = key: exclusive scan over the chunks
            int base = 0, last = -1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cn = s_cnt[c][t], lp = s_lastp[c][t];
                s_cnt[c][t] = (short)base; s_lastp[c][t] = (short)last;
                base += cn; if (cn) last = lp;
            }
        }
    }
    // (3) in the shadow of the hand-offs: the sparse table of d, every element's rank inside the tile and the range maximum since its previous same-key position
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        lds_barrier();
        if (SCAN && l == 1 && t == 0) s_failed = 0;         // (the top level's words are free from here on: the suffix maxima it carried have been read)
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * BLOCK + t;
            const int wq = 1 << (2 * (l - 1));
            int m = s_tbl[l - 1][i];
            if (i - wq >= 0) m = max(m, s_tbl[l - 1][i - wq]);
            if (i - 2 * wq >= 0) m = max(m, s_tbl[l - 1][i - 2 * wq]);
            if (i - 3 * wq >= 0) m = max(m, s_tbl[l - 1][i - 3 * wq]);
            if (l < NL - 1 || i >= TOPFREE) s_tbl[l][i] = m;
        }
    }
    lds_barrier();
    SK1_STAMP(5);
    int rloc[EPT], pp[EPT], rmx[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        rloc[r] = 0; pp[r] = -1; rmx[r] = 0;
        if (key[r] < 0) continue;
        const int l = r * BLOCK + tp, c = r * 4 + wv, ky = key[r];
        rloc[r] = s_cnt[c][ky] + rk[r];
        const int p = (pl[r] >= 0) ? pl[r] : s_lastp[c][ky];
        pp[r] = p;
        const int len = l - p;                              // range max of d over (p, l]
        const int lv = min((31 - __clz(len)) >> 1, NL - 1), wq = 1 << (2 * lv);
        rmx[r] = max(max(s_tbl[lv][l], s_tbl[lv][p + wq]), max(s_tbl[lv][len > 2 * wq ? l - wq : l], s_tbl[lv][len > 3 * wq ? l - 2 * wq : l]));
    }
    const int ginc = wave_iscan_sum(tq), linc = wave_iscan_max(tq ? t + 1 : 0);
    const int lexc = lane_shr1(linc, 0);
    lds_barrier();                                          // every query of the sparse table is done: its top level's first words carry the cross-wave scan
    if (lane == 63) { s_gw[wv] = ginc; s_lw[wv] = linc; }
    SK1_STAMP(6);
    // (3) look-back, level 1: the tiles before this one in its group (the group's last tile has done it above)
    int qc = 0, qt = 0;
    if constexpr (SCAN) {
        // the scanner of this tile's group (same XCD) delivers the final prefix: one lane per wave watches, then the wave's 64 granules
        const unsigned long long *ps = g.scanl + (size_t)w * SKK + t;
        const unsigned want21 = (g.tag & 2047u) << 21;
        unsigned long long fv = 0;
        int spins = 0; unsigned long long t0w = 0;
#ifndef PBWT_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);                      // a waiting tile yields the SIMD to the scanners and aggregators it waits for (and to the tiles still at their tables)
#endif
        for (;;) {
            unsigned long long gv = 0;
            if (lane == 0) gv = __builtin_nontemporal_load(ps);
            if (sk1_tagdiff((unsigned long long)__builtin_amdgcn_readfirstlane((int)(gv >> 32)) << 32, want21) < (1u << 21)) {
                fv = __builtin_nontemporal_load(ps);
                if (__all(sk1_tagdiff(fv, want21) < (1u << 21))) break;
            }
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 63) == 0) {
                const unsigned long long now = wall_clock64();
                if (t0w == 0) t0w = now;
                if (now - t0w > 200000000ULL || __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicCAS(g.err, 0, 11); ok = false; break; }
            }
        }
#ifndef PBWT_NO_SETPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        qc = sk1_cnt(fv); qt = sk1_tail(fv);                // (pc, pt stay the identity: the fold below gives {count, count ? tail : -1})
        SK1_STAMP(2); SK1_STAMP(3);
    } else
    if (BOTH && g.nfold && w - first <= 16 && grp <= 16) {
        // with folders the aggregates are out by the time a 512-position tile has its tables: both levels in ONE round trip (100 k: 1.10 against 1.155 us/site alone,
        // 1.26 against 1.33 beside the consumers; 256-position tiles are behind their tables too early for it: 50 k 1.04 against 0.995 — profiles/r05_onepass.txt, r5j)
        ok = sk1_fold_both<16>(g.rows + (size_t)first * SKK, w - first, g.grows, grp, g.tag, pc, pt, qc, qt, g.err, 11);
        SK1_STAMP(2); SK1_STAMP(3);
    } else {
        if (!selffold) ok = sk1_fold_rows<16>(g.rows + (size_t)first * SKK, w - first, g.tag, pc, pt, g.err, 11);
        SK1_STAMP(2);
        // level 2: the groups before this tile's
        ok = ok && sk1_fold_rows<16>(g.grows, grp, g.tag, qc, qt, g.err, 11);
        SK1_STAMP(3);
    }
    if (!ok) s_failed = 1;
    const int bq = qc + pc, cq = pc ? pt : (qc ? max(qt, pt) : -1);      // keys before the tile, carry (-1: no earlier occurrence)
    g.scan[(size_t)w * SKK + t] = make_int2(bq, cq);       // kept for the fill
    lds_barrier();
    if (s_failed) return;                                   // a bounded wait ran out: nothing is scattered from tables that are not there
    int Gq = ginc - tq, lq = lexc;
    for (int x = 0; x < wv; ++x) { Gq += s_gw[x]; lq = max(lq, s_lw[x]); }
    lq -= 1;
    s_base[t] = Gq + bq;
    s_ext[t] = (cq >= 0) ? (cq | SK_EFLAG) : (lq >= 0 ? g.k + 1 + (31 - __clz(t ^ lq)) : 0);
    lds_barrier();
    // (4) what is left behind the look-back: one addition and one table word per element, and the scatter
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        if (key[r] < 0) continue;
        const int ky = key[r];
        int dd;
        if (pp[r] >= 0) dd = rmx[r];
        else { const int ex = s_ext[ky]; dd = (ex & SK_EFLAG) ? max(ex & ~SK_EFLAG, rmx[r]) : ex; }
        const int pos = s_base[ky] + rloc[r];
        if (pos == 0) dd = g.k + SKB + 1;
        if (g.ycnext) {
            const unsigned tg = g.has_next ? (unsigned)((g.ycnext[pos >> 6] >> (pos & 63)) & 1ULL) : 0u;
            g.a_out[pos] = av[r] | (int)(tg << 31);
            g.d_out[pos] = dd;
        } else {
            g.a_out[pos] = av[r] | (int)((nk[r] & 1u) << 31);
            g.d_out[pos] = dd;
            g.keys_out[pos] = (unsigned char)nk[r];
        }
    }
    if (w == g.Wtot - 1 && t == 0) g.d_out[g.M] = g.k + SKB + 1;
    SK1_STAMP(4);
#undef SK1_STAMP
}
// BOTH (with folders): a tile polls the two levels of its look-back together
template <int EPT, bool BOTH = false, bool MERGED = (EPT == 1), bool SCAN = false>
__global__ __launch_bounds__(BLOCK) void skel_onepass_kernel(SkArgs g) { skel_onepass_body<EPT, BOTH, MERGED, SCAN>(g); }
// the scanner form: eight workgroups per CU (64 VGPRs), so that a 1 M-haplotype launch (1 954 tiles + 70 in front) is resident at once on an idle chip
template <int EPT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void skel_onepass_scan_kernel(SkArgs g) { skel_onepass_body<EPT, false, true, true>(g); }
template <int EPT, bool BOTH = false>
__global__ __launch_bounds__(BLOCK) void skel_onepass_many_kernel(const SkArgs *args) { const SkArgs g = args[blockIdx.y]; skel_onepass_body<EPT, BOTH>(g); }

// PERSISTENT chain of a small panel (<= TR tiles: the two-launch regime): ALL rounds of a batch in ONE launch, hist and rank of every
// round separated by barriers over the launch's <= 128 co-resident workgroups instead of by kernel boundaries.  Such a barrier costs MORE
// than a boundary (>= 4 us against 1.5-2.5), so this is not how a lone small panel runs fastest — it is how a small
// panel's chain stays OFF the launch stream of a wide one: the query cursor of matchSequencesSweep (10 000 haplotypes beside a panel of
// 10^6) costs 128 dependent launches per 512-site batch, a third of what bounds that job; here it costs one.
// rounds[s] = the arguments of round s (device memory).  The counter only grows: barrier i of this launch waits for base + (i+1) * gridDim.x.
__device__ __forceinline__ void skel_grid_barrier(unsigned *counter, unsigned target, int *err) {
    __syncthreads();                                        // every wave's stores are out (vmcnt(0)) before thread 0 releases them
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 25)) { atomicExch(err, 8); break; }
        }
    }
    __syncthreads();
}
template <int EPT, int TR>
__global__ __launch_bounds__(BLOCK) void skel_persist_kernel(const SkArgs *rounds, int nr, unsigned *counter, unsigned base, int *err) {
    const unsigned nwg = gridDim.x;
    unsigned target = base;
    for (int s = 0; s < nr; ++s) {
        const SkArgs g = rounds[s];
        skel_hist_body<EPT, false>(g);
        target += nwg; skel_grid_barrier(counter, target, err);          // every tile's row is in the table
        skel_rank_body<EPT, TR, false, false>(g, nullptr);
        target += nwg; skel_grid_barrier(counter, target, err);          // the new state (a, d, keys) is complete
    }
}

// TEAM-PERSISTENT chain (round 5; the probe is tools/latprobe4.hip section 5, profiles/r04_latprobe4.txt): ALL rounds of a batch in
// ONE launch, hist -> per-key scan -> rank of every round separated by barriers among the workgroups of ONE XCD instead of by kernel boundaries.  A
// dependent launch costs 3.5-3.9 us whatever it carries; inside one XCD the same dependency is a flag-word barrier through that XCD's L2 — no far
// atomics, no fences: arrive = plain store of the round number into the member's own flag word (it lands in the XCD's L2, the coherence point of
// every CU of the team), poll = sc1 loads of all K words by one wave (they bypass the CU's L1 and are served by that L2), payload = plain stores read
// back with nontemporal loads (SKM_TEAM) — 0.45 us per barrier, 0.75-0.95 us with a 256 KB hand-off in the probe.
// Launch: 8 (K + slack) workgroups; a workgroup reads HW_REG_XCC_ID, leaves unless its XCD carries a panel (panel p lives on XCD x0 + p: eight
// chromosomes of one cohort side by side, pbwtIO.c:477-483 once per panel), takes a ticket in its XCD's team and leaves if the team is full.  Member m
// takes the tiles m, m + K, ... of its panel and the key groups m, m + K, ... of the scan.  Placement is read, not assumed: a team that does not fill
// (the observed "block b runs on XCD b mod 8" not holding) runs into the barrier's bounded wait and fails the pass (device error 10).
// rounds[s * P + p] = the arguments of round s of panel p.  The flag words only grow: barrier i of this launch carries the round number round0 + i + 1.
constexpr int TEAM_MAXK = 256;                              // members per team (flag words per XCD)
constexpr int TEAM_TICKETS = 64;                            // words in front of the flags: [x] = tickets taken on XCD x in this launch (zeroed before every launch)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }     // HW_REG_XCC_ID[3:0]
// where the workgroups of a launch on this stream land: out[0] |= 1 << XCD of every workgroup, out[1 + b] = XCD of workgroup b (64 workgroups).  The engine
// reads it once at creation: the team-persistent chain needs XCD x0 + p to exist for every panel p, the scanner form of the one-launch round needs "workgroup b
// runs on XCD b mod 8" (its groups hand off through one XCD's L2 with plain stores)
__global__ __launch_bounds__(64) void xcd_probe_kernel(unsigned *out) {
    if (threadIdx.x == 0) { const unsigned x = (unsigned)xcc_id(); atomicOr(out, 1u << x); out[1 + (blockIdx.x & 63)] = x; }
}

__device__ __forceinline__ bool team_barrier(unsigned *flags, int m, int K, unsigned round, int *err, int *s_abort) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave's stores acknowledged by the L2 ...
    __syncthreads();                                        // ... before the member's flag says so
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + m, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_s_setprio(0);                      // a polling wave yields to whatever shares its SIMD
        const unsigned long long t0 = wall_clock64();       // 100 MHz
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (int i = threadIdx.x; i < K; i += 64) ok &= (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - round) >= 0;
            if (__all(ok)) break;
            // bounded (4 s of wall clock): members wait here for the last of them to be placed beside the consumer kernels, never for ever
            if ((++spins & 255) == 0 && (wall_clock64() - t0 > 400000000ULL || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (threadIdx.x == 0) { atomicCAS(err, 0, 10); *s_abort = 1; }
                break;
            }
        }
        __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    return *s_abort == 0;
}

// prof (PBWTAMD_TEAM_PROF=1, nullptr otherwise): member 0 of panel 0 stamps the wall clock (100 MHz) before and after every barrier of every round: [round][8]
template <int EPT, int KPW, int TPL>
__global__ __launch_bounds__(BLOCK) void skel_team_kernel(const SkArgs *rounds, int nr, int P, int x0, int K, unsigned *ctl, unsigned round0, int *err, unsigned long long *prof) {
    __shared__ int s_m, s_abort;
    const int x = xcc_id(), p = x - x0;
    if (p < 0 || p >= P) return;
    if (threadIdx.x == 0) { s_m = (int)__hip_atomic_fetch_add(ctl + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 0; }
    __syncthreads();
    const int m = s_m;
    if (m >= K) return;
    unsigned *flags = ctl + TEAM_TICKETS + (size_t)x * TEAM_MAXK;
    unsigned round = round0;
    const bool stamp = prof && p == 0 && m == 0 && threadIdx.x == 0;
#define TEAM_STAMP(i) do { if (stamp) prof[(size_t)s * 8 + (i)] = wall_clock64(); } while (0)
    for (int s = 0; s < nr; ++s) {
        const SkArgs g = rounds[(size_t)s * P + p];
        TEAM_STAMP(0);
        for (int w = m; w < g.W; w += K) { skel_hist_body<EPT, false, SKM_TEAM>(g, w); lds_barrier(); }
        TEAM_STAMP(1);
        if (!team_barrier(flags, m, K, ++round, err, &s_abort)) return;            // every tile's row is in the table
        TEAM_STAMP(2);
        {
            Sk2Args k2; k2.tbl = g.tbl; k2.scan = g.scan; k2.total = g.total; k2.W = g.W;
            for (int kb = m; kb < SKK / KPW; kb += K) { skel_k2_body<KPW, TPL, SKM_TEAM>(k2, kb); __syncthreads(); }
        }
        TEAM_STAMP(3);
        if (!team_barrier(flags, m, K, ++round, err, &s_abort)) return;            // every key's prefixes and total are out
        TEAM_STAMP(4);
        for (int w = m; w < g.W; w += K) { skel_rank_body<EPT, 0, true, false, SKM_TEAM>(g, nullptr, w); lds_barrier(); }
        TEAM_STAMP(5);
        if (!team_barrier(flags, m, K, ++round, err, &s_abort)) return;            // the new state (a, d, keys) is complete
        TEAM_STAMP(6);
    }
#undef TEAM_STAMP
}

// MANY PANELS PER LAUNCH (pbwtamd_pass_advance_many): P independent panels of the same width — chromosomes side by side — advance through the
// same round in the same three (two) launches: blockIdx.y = panel, args[panel] = that panel's arguments for the round (device memory, the
// whole batch uploaded at once).  Below ~250 k haplotypes a chain launch costs its 3-4 us whatever runs inside it, so P panels per launch
// cost little more than one.
template <int EPT>
__global__ __launch_bounds__(BLOCK) void skel_hist_many_kernel(const SkArgs *args) { const SkArgs g = args[blockIdx.y]; skel_hist_body<EPT, false>(g); }
template <int KPW, int TPL>
__global__ __launch_bounds__(KPW * 64) void skel_k2_many_kernel(const SkArgs *args) {
    const SkArgs g = args[blockIdx.y];
    Sk2Args k; k.tbl = g.tbl; k.scan = g.scan; k.total = g.total; k.W = g.W;
    skel_k2_body<KPW, TPL>(k);
}
template <int EPT, int TR, bool R4 = false>
__global__ __launch_bounds__(BLOCK) void skel_rank_many_kernel(const SkArgs *args) { const SkArgs g = args[blockIdx.y]; skel_rank_body<EPT, TR, R4, false>(g, nullptr); }

// READ SIDE: the columns arrive in PBWT order (y_k by position), so the 8-bit key of position i of the
// state before site k follows the LF-mapping through the 8 columns: bit j = y_{k+j}[p_j], p_0 = i,
// p_{j+1} = y ? c + p_j - u(p_j) : u(p_j) with u = zeros before p_j (rank directory + popcount).  It
// depends on the columns only, not on a[]: all rounds of a batch at once.  grid (tiles, rounds).
__global__ __launch_bounds__(BLOCK) void skel_keys_sorted_kernel(const unsigned long long *ycols, int wpc64, const int *rankdir, int M,
                                                                unsigned char *keys, size_t strideK) {
    const int r = blockIdx.y, i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= M) return;
    int pos = i;
    unsigned key = 0;
#pragma unroll
    for (int j = 0; j < SKB; ++j) {
        const int site = SKB * r + j;
        const unsigned long long w = ycols[(size_t)site * wpc64 + (pos >> 6)];
        const int *rd = rankdir + (size_t)site * (wpc64 + 1);
        const unsigned bit = (unsigned)((w >> (pos & 63)) & 1ULL);
        key |= bit << j;
        const int u = rd[pos >> 6] + ((pos & 63) - __popcll(w & ((1ULL << (pos & 63)) - 1ULL)));
        pos = bit ? rd[wpc64] + pos - u : u;
    }
    keys[(size_t)r * strideK + i] = (unsigned char)key;
}

// read side: tag slot 0 of a batch with its column (by position)
__global__ void skel_tag_sorted_kernel(int *a, const unsigned long long *yc, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) a[i] = (a[i] & AMASK) | (int)((unsigned)((yc[i >> 6] >> (i & 63)) & 1ULL) << 31);
}

// keys (and tags) of a state from the transposed panel: start of a batch
__global__ void skel_keys_kernel(int *a, const unsigned char *kb, int M, unsigned char *keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) { const int v = a[i] & AMASK; const unsigned kk = kb[v]; a[i] = v | (int)((kk & 1u) << 31); keys[i] = (unsigned char)kk; }
}

}  // namespace pbwtk
