// pbwt_k_sweep.h — matchMaximalWithin / matchLongWithin sweeps over ring slots (pbwtMatch.c:85-142): record sinks, the streaming histogram sweep, the residual sweep behind the fused fill.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// matchMaximalWithin sweep (pbwtMatch.c:115-142) over ring slots holding (a_k tagged with y_k, d_k).
// One thread per position i; grid (tiles, sites).  MODE 0: count reports per block; 1: emit
// records at precomputed block offsets; 2: histogram (pbwtMatch.c:130-131).
// `final_site` = index in this batch of the k == N state (all positions report, y ignored) or -1.
struct SweepArgs {
    const int *A; const int *D; size_t strideA, strideD;
    int M, kbase, final_site;
    unsigned long long *blockCount;      // [sites*tiles]   MODE 0 out / MODE 1 in (exclusive offsets)
    int4 *recs;                          // MODE 1
    unsigned long long *hist; int histlen;  // MODE 2
    int *err;
    unsigned long long *ycols; int wpc64;   // MODE 2, optional: also emit the sorted bit column of each site (what pack3 encodes)
#ifdef PBWTAMD_MEASURE
    int dbg;                                // measurement builds only (results WRONG): 1 = no histogram atomics, 2 = no walks either
#endif
    int nvb;                                // 256-position blocks per site
    unsigned long long *hist_rep;           // streaming form: HIST_REP copies of the first HIST_LBINS bins, folded into hist by hist_fold_kernel
    int iters;                              // streaming form: 1024-position groups per workgroup
    int clip;                               // P16: lengths from clip on are escapes (P16_ESC; smaller in tests)
    const unsigned short *P16; size_t stride16;  // streaming form, 16-bit hand-off (skel_fillseq_kernel<.., 3>): slots of L | y << 15, L = site + 1 - d, P16_ESC = "read d from D"
};
// Same-address global atomics serialise chip-wide (~12 ns each): a panel whose matches all have similar lengths (iid: every
// report lands in ~30 bins) would spend seconds there.  So the short lengths are counted in LDS per workgroup first and
// flushed to one of HIST_REP replicas of the low bins; long lengths (spread over many bins) go straight to hist.
constexpr int HIST_LBINS = 2048, HIST_REP = 32;
#ifdef PBWTAMD_WALKSTAT                                     // (its own switch: the counters' atomics slow a measurement build fivefold)
__device__ unsigned long long g_walkstat[16];               // [0] LDS walks [1] their 256-position steps [2] memory walks [3] their steps [4..11] LDS walks by steps: 1, 2, 3-4, 5-8, 9-16, 17-32, > 32 ; [12] pending up [13] pending down
#define WALKSTAT(i, n) do { if (lane_id() == 0) atomicAdd(&g_walkstat[i], (unsigned long long)(n)); } while (0)
#else
#define WALKSTAT(i, n) do { } while (0)
#endif
__global__ void hist_fold_kernel(unsigned long long *hist, unsigned long long *rep, int histlen) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= HIST_LBINS) return;
    unsigned long long s = 0;
    for (int r = 0; r < HIST_REP; ++r) { s += rep[(size_t)r * HIST_LBINS + b]; rep[(size_t)r * HIST_LBINS + b] = 0; }
    if (s && b < histlen) hist[b] += s;
}

// wave-cooperative walk: from position `from` in direction `dir` (-1 up, +1 down) find the first
// position p with d[p + off] > thr (the block boundary; `stop` = p) or, unless `fin`, with allele
// == yi (then the match extends: returns true = skip).  64 positions per step via ballot.
// All 64 lanes call this with wave-uniform arguments.
template <bool PACKED>
__device__ __forceinline__ bool coop_walk(const int *a, const int *d, int from, int dir, int thr, unsigned yi, bool fin, int M, int &stop) {
    const int lane = lane_id();
    for (;;) {
        const int p = from + dir * lane;                   // candidate neighbour index (m or n of the reference loop)
        // up:   loop test d[m+1] <= thr  with m = p  -> boundary when d[p+1] > thr ; y test on y[p]
        // down: loop test d[n]   <= thr  with n = p  -> boundary when d[p]   > thr ; y test on y[p]
        const int di = (dir < 0) ? p + 1 : p;
        const bool inb = (di >= 0) && (di <= M);
        const bool bound = inb ? ((PACKED ? (d[di] & 0x7fffffff) : d[di]) > thr) : true;
        const bool same = (!bound && !fin && p >= 0 && p < M) ? (((unsigned)(PACKED ? d[p] : a[p]) >> 31) == yi) : false;
        const unsigned long long mb = __ballot(bound), ms = __ballot(same);
        const unsigned long long any = mb | ms;
        if (any) {
            const int first = __ffsll((long long)any) - 1;
            if ((ms >> first) & 1ULL) return true;         // same allele inside the block: i is not reported
            stop = from + dir * first;                     // first index that fails the divergence test
            return false;
        }
        from += dir * 64;
    }
}

// PACKED (MODE 2 only): the slots hold d | y << 31 in D and A is not read (what skel_fill_kernel writes
// when no consumer needs the haplotype ids): half the bytes of the sweep.
template <int MODE, bool PACKED = false, int ITC = 1>
__global__ __launch_bounds__(BLOCK) void sweep_within_kernel(SweepArgs g) {
    __shared__ unsigned long long s_w[WAVES];
    auto DV = [&](const int *dd, int x) -> int { return PACKED ? (dd[x] & 0x7fffffff) : dd[x]; };
    auto YV = [&](const int *aa, const int *dd, int x) -> unsigned { return (unsigned)(PACKED ? dd[x] : aa[x]) >> 31; };
    const int site = blockIdx.y, k = g.kbase + site;
    const bool fin = (site == g.final_site);
    const int *a = g.A + (size_t)site * g.strideA;
    const int *d = g.D + (size_t)site * g.strideD;
    const int M = g.M;
    const int lane = lane_id();
    // MODE 2 walks g.iters consecutive 256-position blocks per workgroup (at M = 1M one block per workgroup is 2M workgroups
    // per batch: dispatch-bound); the record modes keep one block per workgroup (their offsets are per block)
    constexpr int IT = (MODE == 2) ? ITC : 1;
    // every entry is handled as one word d | y << 31 (the packed slots hold exactly that; otherwise d and the tag of a are
    // merged on load).  The own word and its three neighbours of all IT blocks are requested up front: the first step of
    // both scans and the stop test of the second are then decided from registers, and 4 x IT loads are in flight per lane.
    auto WD = [&](int x) -> int { return PACKED ? __builtin_nontemporal_load(d + x) : (__builtin_nontemporal_load(d + x) | (__builtin_nontemporal_load(a + x) & (int)0x80000000)); };
    int pre_m[IT], pre_0[IT], pre_1[IT], pre_2[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int ii = (blockIdx.x * IT + it) * BLOCK + threadIdx.x;
        const bool in = ii < M;
        pre_m[it] = (in && ii > 0) ? WD(ii - 1) : 0; pre_0[it] = in ? WD(ii) : 0;
        pre_1[it] = in ? WD(ii + 1) : 0; pre_2[it] = (in && ii + 2 <= M) ? WD(ii + 2) : 0;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
    const int vb = blockIdx.x * IT + it;
    if (vb >= g.nvb) break;
    const int i = vb * BLOCK + threadIdx.x;
    int m = i - 1, n = i + 1, di = 0, dn = 0;
    unsigned yi = 0;
    bool rep = false;
    // scalar prefix of the reference's two scans (pbwtMatch.c:124-129), a few steps per lane; the rare
    // long walks (a rare allele beside a long run of the other one) are finished wave-cooperatively
    constexpr int BUDGET = 4;
    bool needUp = false, needDown = false;
    if (i < M) {
        di = pre_0[it] & 0x7fffffff; yi = (unsigned)pre_0[it] >> 31; dn = pre_1[it] & 0x7fffffff;
        rep = true;
#ifdef PBWTAMD_MEASURE
        if (g.dbg == 2) { /* loads only */ } else
#endif
        if (di <= dn) {                                     // while (d[m+1] <= d[i]) if (y[m--] == y[i]) skip   (pbwtMatch.c:124-126)
            int steps = 0, wcur = pre_0[it];                // wcur = the word at m+1
            for (;;) {
                if ((wcur & 0x7fffffff) > di) break;
                const int wm = (m == i - 1) ? pre_m[it] : WD(m);
                if (!fin && ((unsigned)wm >> 31) == yi) { rep = false; break; }
                --m; wcur = wm;
                if (++steps == BUDGET) { needUp = true; break; }
            }
        }
        if (rep && !needUp && di >= dn) {                   // while (d[n] <= d[i+1]) if (y[n++] == y[i]) skip    (pbwtMatch.c:127-129)
            int steps = 0, wn = pre_1[it];
            for (;;) {
                if ((wn & 0x7fffffff) > dn) break;
                if (!fin && ((unsigned)wn >> 31) == yi) { rep = false; break; }
                ++n;
                if (++steps == BUDGET) { needDown = true; break; }
                wn = (n == i + 2) ? pre_2[it] : WD(n);
            }
        }
    }
    if constexpr (MODE == 2) {
        if (g.ycols) {                                      // the tags of this site as a sorted bit column (saves pack3 a pass over A)
            const unsigned long long mk = __ballot(i < M && yi);
            const int wd = vb * WAVES + wave_id();
            unsigned long long *yc = g.ycols + (size_t)site * g.wpc64;
            if (lane == 0 && wd < g.wpc64) yc[wd] = mk;
            if (vb == g.nvb - 1) for (int x = g.nvb * WAVES + threadIdx.x; x < g.wpc64; x += BLOCK) yc[x] = 0ULL;
        }
    }
    // finish long upward walks, one lane at a time, all 64 lanes scanning
    for (unsigned long long pend = __ballot(needUp); pend; pend &= pend - 1) {
        const int src = __ffsll((long long)pend) - 1;
        const int from = __builtin_amdgcn_readlane(m, src), thr = __builtin_amdgcn_readlane(di, src);
        const unsigned yy = (unsigned)__builtin_amdgcn_readlane((int)yi, src);
        int stop = 0;
        const bool skip = coop_walk<PACKED>(a, d, from, -1, thr, yy, fin, M, stop);
        if (lane == src) { if (skip) rep = false; else m = stop; }
    }
    // lanes whose upward walk was long still owe the downward scan
    if (needUp && rep && di >= dn) {
        int steps = 0;
        while (DV(d, n) <= dn) {
            if (!fin && YV(a, d, n) == yi) { rep = false; break; }
            ++n;
            if (++steps == BUDGET) { needDown = true; break; }
        }
    }
    for (unsigned long long pend = __ballot(needDown && rep); pend; pend &= pend - 1) {
        const int src = __ffsll((long long)pend) - 1;
        const int from = __builtin_amdgcn_readlane(n, src), thr = __builtin_amdgcn_readlane(dn, src);
        const unsigned yy = (unsigned)__builtin_amdgcn_readlane((int)yi, src);
        int stop = 0;
        const bool skip = coop_walk<PACKED>(a, d, from, +1, thr, yy, fin, M, stop);
        if (lane == src) { if (skip) rep = false; else n = stop; }
    }
    if (MODE == 2) {
        if (rep) {
            const int len = (di < dn) ? k - di : k - dn;
#ifdef PBWTAMD_MEASURE
            if (g.dbg) { if (len == -12345) g.hist[0] = 1; } else
#endif
            if (len >= 0 && len < g.histlen) atomicAdd(g.hist + len, 1ULL); else atomicExch(g.err, 1);
        }
        continue;
    }
    const unsigned long long cnt = rep ? (unsigned long long)((i - 1 - m) + (n - 1 - i)) : 0ULL;
    // block exclusive scan of cnt
    unsigned long long inc = cnt;
    for (int o = 1; o < 64; o <<= 1) { unsigned long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_w[wave_id()] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
    const size_t bidx = (size_t)site * g.nvb + vb;
    if (MODE == 0) { if (threadIdx.x == 0) g.blockCount[bidx] = tot; return; }
    if (rep && cnt) {
        int4 *out = g.recs + g.blockCount[bidx] + pre + (inc - cnt);
        const int ai = a[i] & AMASK;
        for (int jj = m + 1; jj < i; ++jj) *out++ = make_int4(ai, a[jj] & AMASK, di, k);
        for (int jj = i + 1; jj < n; ++jj) *out++ = make_int4(ai, a[jj] & AMASK, dn, k);
    }
    }
}

// as coop_walk, 256 positions per step (four independent loads per lane in flight): the long walks of the histogram
// sweep — a rare allele beside a block of thousands of identical haplotypes carrying the other one — are chains of
// dependent round trips, so fewer, wider steps.  Only the decision is returned (the histogram needs no stop index).
template <bool PACKED, bool P16 = false>
__device__ __forceinline__ bool coop_walk4(const int *a, const int *d, int from, int dir, int thr, unsigned yi, int M, const unsigned short *p16 = nullptr, int kp1 = 0) {
    const int lane = lane_id();
    WALKSTAT(2, 1);
    for (;; from += dir * 256) {
        WALKSTAT(3, 1);
        int wd[4], wy[4];
        if constexpr (P16) {                                // all eight raw words in flight first; escapes (rare) are fetched behind one wave-uniform test
            unsigned rd[4], ry[4]; bool esc = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = from + dir * (lane + 64 * j), di = (dir < 0) ? p + 1 : p;
                rd[j] = __builtin_nontemporal_load(p16 + min(max(di, 0), M));
                ry[j] = (dir < 0) ? (unsigned)__builtin_nontemporal_load(p16 + min(max(p, 0), M)) : rd[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) esc = esc || (rd[j] & 0x7fffu) == 0x7fffu;      // (only the divergences matter; of ry only the allele bit is used)
            const bool anyesc = __ballot(esc) != 0ULL;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = from + dir * (lane + 64 * j), di = (dir < 0) ? p + 1 : p;
                const bool inb = (di >= 0) && (di <= M);
                const int dw = anyesc ? p16_word(rd[j], kp1, d, min(max(di, 0), M)) : ((kp1 - (int)(rd[j] & 0x7fffu)) | (int)((rd[j] >> 15) << 31));
                wd[j] = inb ? dw : 0x7fffffff;
                wy[j] = (p >= 0 && p < M) ? (int)((ry[j] >> 15) << 31) : (int)((yi ^ 1u) << 31);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = from + dir * (lane + 64 * j);
            const int di = (dir < 0) ? p + 1 : p;           // the divergence tested for candidate p (see coop_walk)
            const bool inb = (di >= 0) && (di <= M);
            wd[j] = inb ? __builtin_nontemporal_load(d + di) : 0x7fffffff;
            wy[j] = (p >= 0 && p < M) ? (PACKED ? ((dir < 0) ? __builtin_nontemporal_load(d + p) : wd[j]) : __builtin_nontemporal_load(a + p)) : (int)((yi ^ 1u) << 31);
        }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool bound = (wd[j] & 0x7fffffff) > thr;
            const bool same = !bound && (((unsigned)wy[j] >> 31) == yi);
            const unsigned long long mb = __ballot(bound), ms = __ballot(same), any = mb | ms;
            if (any) return (ms >> (__ffsll((long long)any) - 1)) & 1ULL;      // the first event in walking order decides: same allele = skip
        }
    }
}

// matchMaximalWithin, histogram sink (pbwtMatch.c:115-131 with matchLengthHist set): the streaming form.  A wave owns 256
// consecutive positions as four 64-position chunks (one coalesced load each, neighbours by DPP), and almost every position
// is decided from its own word and its two neighbours: with b = y[i],
//     d[i] <= d[i+1] and y[i-1] == b   -> the upward scan meets b at its first step: not reported
//     d[i] >= d[i+1] and y[i+1] == b   -> the downward scan does: not reported
// What is left are run boundaries of the allele column whose scan has to go on (pbwtMatch.c:124-129: until a divergence
// above the threshold ends the block, or the same allele turns up).  Those few are resolved wave-cooperatively: first
// inside the wave's own 256 words with ballots (no memory access), then 256 positions per step through memory.
// Emits the site's sorted bit column as a by-product (one ballot per chunk) when ycols is set.
// YCIN (round 4; with PACKED): the sites' sorted allele columns are an INPUT — the fill emitted them (skel_fillseq_kernel<.., YC>) — and the sweep
// reads THOSE first, one bit per position: a wave takes 64 column words = 16 groups of 256 positions, finds the groups that are not y-uniform
// (all four words 0 or all ~0, and the neighbouring bit on either side the same) from ballots of the words, and loads d | y only for them — on a
// founder-mosaic panel a quarter of the groups.  Everything a loaded group goes through is the code below, unchanged.
// P16 (round 4; with PACKED): the slots are the 16-bit ring of skel_fillseq_kernel<.., 3> — half the bytes; a group's words are converted to the
// d | y << 31 form when the group is processed (after the y-uniform test, which needs the allele bits only).
template <bool PACKED, bool YCIN = false, bool P16 = false>
__global__ __launch_bounds__(BLOCK) void sweep_hist_kernel(SweepArgs g) {
    constexpr int CH = 4;
    // (P16: two 16-bit counters per word — a workgroup has at most 8192 positions, so no bin can reach 65 536 — 4 KB instead of 8: the LDS copy
    // of the range below must leave the chain's workgroups room beside this kernel)
    __shared__ unsigned s_hist[P16 ? HIST_LBINS / 2 : HIST_LBINS];
    // P16: the workgroup's whole range — iters x 4 groups = up to 8192 consecutive positions, 16 KB — is staged in LDS first (16-byte loads, all in flight
    // at once), and the scans that leave a wave's own 256 words WALK THE LDS COPY: the walks through memory were 40 % of this kernel (chains of
    // dependent round trips of ~2 us, one pending position after the other); only a walk that leaves the workgroup's range still goes to memory.
    constexpr int RANGE = P16 ? 8192 : 8;
    __shared__ __attribute__((aligned(16))) unsigned short s_w[RANGE + 8];
    const int site = blockIdx.y, k = g.kbase + site;
    const bool fin = (site == g.final_site);
    const int *a = g.A + (size_t)site * g.strideA;
    const int *d = g.D + (size_t)site * g.strideD;
    const int M = g.M, lane = lane_id();
    const unsigned short *p16 = P16 ? g.P16 + (size_t)site * g.stride16 : nullptr;
    const int kp1 = k + 1;
    for (int x = threadIdx.x; x < (P16 ? HIST_LBINS / 2 : HIST_LBINS); x += BLOCK) s_hist[x] = 0;
    const int r0 = blockIdx.x * g.iters * WAVES * (64 * CH), rn = min(g.iters * WAVES * (64 * CH), RANGE);       // the workgroup's positions [r0, r0 + rn)
    int edgeL = 0, edgeR = 0;                               // P16: the words just outside the range
    if constexpr (P16) {
        for (int x = threadIdx.x * 8; x < rn; x += BLOCK * 8)
            if (r0 + x <= M) *reinterpret_cast<uint4 *>(s_w + x) = *reinterpret_cast<const uint4 *>(p16 + r0 + x);     // (a slot is followed by another slot or the ring's padding)
        edgeL = (int)__builtin_nontemporal_load(p16 + max(r0 - 1, 0)); edgeR = (int)__builtin_nontemporal_load(p16 + min(r0 + rn, M));
    }
    __syncthreads();
    // branch-free loads: every address is clamped into [0, M] (index M holds the sentinel d[M]); words of positions beyond M
    // are never used as anything but a right neighbour of an invalid position
    auto WD = [&](int x) -> int {
        const int xc = min(max(x, 0), M);
        if constexpr (P16) return (int)__builtin_nontemporal_load(p16 + xc);      // raw: L | y << 15 (converted in process())
        return PACKED ? __builtin_nontemporal_load(d + xc) : (__builtin_nontemporal_load(d + xc) | (__builtin_nontemporal_load(a + min(xc, M - 1)) & (int)0x80000000));
    };
    // the words of a group (own 4 chunks + the two halo words, wave-uniform addresses) are requested TWO iterations ahead of their use, in two
    // register sets: the kernel is bound by the bytes it keeps in flight (8 waves per SIMD x 1.5 KB per wave and group), not by issue — with one
    // group in flight it read 2.5 TB/s whether or not the y-uniform fast path below removed most of its instructions
    struct Grp { int w[CH]; int hl, hr; };
    // P16: a lane loads PAIRS — two dwords = positions wb + 128 c + 2 lane, + 1 (c = 0, 1) — half the load instructions of the 32-bit form; the allele
    // test of the y-uniform path works on the pairs as they are, and only a group that is looked at redistributes them (one ds_bpermute per chunk).
    // Nothing is clamped: a slot is followed by another slot or the ring's padding, and words beyond position M are never used (see below).
    auto request = [&](Grp &q, int wvq) {                   // wvq = index of the 256-position group
        const int wb = wvq * (64 * CH);
#pragma unroll
        for (int c = 0; c < CH; ++c) q.w[c] = WD(wb + 64 * c + lane);
        q.hl = WD(wb - 1);
        q.hr = WD(wb + 64 * CH);
    };
    // P16: a wave takes iters CONSECUTIVE groups, so the words of the site's sorted bit column it produces are consecutive too: they collect in a
    // register (lane = word) and leave as ONE store per wave instead of one per y-uniform group and four per other group (§2's 16-cycle rule)
    unsigned long long ycacc = 0ULL; int ycslot = 0;
    // P16: the scans of pbwtMatch.c:124-129 over the LDS copy, 256 candidates per step, in the L domain (d > thr <=> L < kp1 - thr; an escaped L is
    // >= clip, so it never ends a block as long as kp1 - thr <= clip — the caller goes to memory otherwise).  Returns 1 = the same allele
    // turned up (not reported), 2 = the block ended first, 0 = the range ended first: `from` is then the first candidate outside it.
    auto walk_lds = [&](int &from, int dir, int thr, unsigned b) -> int {
        const int Lthr = kp1 - thr;
        int nst = 0; (void)nst;
        auto fin_stat = [&]() { WALKSTAT(0, 1); WALKSTAT(1, nst); WALKSTAT(4 + (nst <= 1 ? 0 : nst == 2 ? 1 : nst <= 4 ? 2 : nst <= 8 ? 3 : nst <= 16 ? 4 : nst <= 32 ? 5 : 6), 1); };
        for (;;) {
            if (dir < 0 ? (from < r0) : (from >= r0 + rn)) { fin_stat(); return 0; }
            ++nst;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = from + dir * (lane + 64 * j), rq = q - r0;
                const bool in = (dir < 0) ? (rq >= 0) : (rq < rn && q <= M);
                // up: candidate q stops the scan when d[q + 1] > thr, skips i when y[q] == b; down: d[q] > thr (d[M] is the sentinel), y[q] == b (q < M)
                const unsigned hd = in ? (unsigned)s_w[(dir < 0) ? rq + 1 : rq] : 0u, hy = (dir < 0) ? (in ? (unsigned)s_w[rq] : 0u) : hd;
                const bool bound = in && (int)(hd & 0x7fffu) < Lthr;
                const bool same = in && !bound && (dir < 0 || q < M) && (hy >> 15) == b;
                const unsigned long long mb = __ballot(bound), ms = __ballot(same), any = mb | ms;
                if (any) { fin_stat(); return ((ms >> (__ffsll((long long)any) - 1)) & 1ULL) ? 1 : 2; }
                if (__ballot(!in)) { from = (dir < 0) ? r0 - 1 : r0 + rn; fin_stat(); return 0; }      // the step crossed the end of the range undecided
            }
            from += dir * 256;
        }
    };
    auto process = [&](int qw0, int qw1, int qw2, int qw3, int qhl, int qhr, int wv) -> bool {      // false: beyond the panel, stop  (the group's words by value: no struct through memory)
    const int qw[CH] = {qw0, qw1, qw2, qw3};
    const int wbase = wv * (64 * CH);
    int w[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) w[c] = qw[c];
    int hl = P16 ? (int)((unsigned)qhl << 16) : qhl, hr = P16 ? (int)((unsigned)qhr << 16) : qhr;      // (P16: the allele bit to the sign position, for the test below)
    if (wbase > M) return false;
    // Y-UNIFORM GROUP: when the wave's 256 positions and their two neighbours all carry the same allele, every position has that allele on
    // the side its scan starts from (d[i] <= d[i+1]: y[i-1]; else d[i] >= d[i+1]: y[i+1]), so nothing reports and nothing scans on
    // (pbwtMatch.c:124-129) — four sign tests instead of ~430 instructions.  On a founder-mosaic panel 3 groups in 4 are like that (most sites
    // carry a rare allele); the group holding position 0 or M and the k == N sweep take the general path.
    if (!YCIN && !fin && wbase > 0 && wbase + 64 * CH < M) {
        bool all0, all1;
        const int h0 = __builtin_amdgcn_readfirstlane(hl), h1 = __builtin_amdgcn_readfirstlane(hr);
        if constexpr (P16) {                                // both allele bits of every pair
            const unsigned long long any1 = __ballot(((w[0] | w[1]) & (int)0x80008000) != 0), any0 = __ballot(((w[0] & w[1]) & (int)0x80008000) != (int)0x80008000);
            all0 = any1 == 0ULL && h0 >= 0 && h1 >= 0; all1 = any0 == 0ULL && h0 < 0 && h1 < 0;
        } else {
            const unsigned long long m0 = __ballot(w[0] < 0), m1 = __ballot(w[1] < 0), m2 = __ballot(w[2] < 0), m3 = __ballot(w[3] < 0);
            all0 = (m0 | m1 | m2 | m3) == 0ULL && h0 >= 0 && h1 >= 0; all1 = (m0 & m1 & m2 & m3) == ~0ULL && h0 < 0 && h1 < 0;
        }
        if (all0 || all1) {
            if constexpr (P16) { if ((lane >> 2) == (ycslot >> 2)) ycacc = all1 ? ~0ULL : 0ULL; }
            else
            if (g.ycols && lane < CH) (g.ycols + (size_t)site * g.wpc64)[wv * CH + lane] = all1 ? ~0ULL : 0ULL;    // wv * CH + 3 < wpc64: the group ends before M
            return true;
        }
    }
    if constexpr (P16) {                                    // pairs -> one position per lane and chunk, then the d | y << 31 words of the group
        unsigned hw[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int v = __builtin_amdgcn_ds_bpermute(((c & 1) * 32 + (lane >> 1)) << 2, qw[c >> 1]);
            hw[c] = (lane & 1) ? ((unsigned)v >> 16) : ((unsigned)v & 0xffffu);
        }
        const unsigned uhl = (unsigned)qhl, uhr = (unsigned)qhr;
        bool esc = (uhl & 0x7fffu) == 0x7fffu || (uhr & 0x7fffu) == 0x7fffu;
#pragma unroll
        for (int c = 0; c < CH; ++c) esc = esc || (hw[c] & 0x7fffu) == 0x7fffu;
        if (__ballot(esc)) {                                // (wave-uniform, rare: a match of 32 767 sites or more) escapes fetch d from the 32-bit slot
#pragma unroll
            for (int c = 0; c < CH; ++c) w[c] = p16_word(hw[c], kp1, d, min(wbase + 64 * c + lane, M));
            hl = p16_word(uhl, kp1, d, max(wbase - 1, 0)); hr = p16_word(uhr, kp1, d, min(wbase + 64 * CH, M));
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) w[c] = (kp1 - (int)(hw[c] & 0x7fffu)) | (int)((hw[c] >> 15) << 31);
            hl = (kp1 - (int)(uhl & 0x7fffu)) | (hl & (int)0x80000000); hr = (kp1 - (int)(uhr & 0x7fffu)) | (hr & (int)0x80000000);
        }
    }
    int dI[CH], dN[CH]; unsigned yI[CH];
    bool pendUp[CH], pendDn[CH], rep[CH];
    unsigned long long mPendUp = 0, mPendDn = 0;             // any pending lane in the wave (per chunk bit sets are re-balloted below)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int p = wbase + 64 * c + lane;
        const int fillL = (c > 0) ? __builtin_amdgcn_readlane(w[c > 0 ? c - 1 : 0], 63) : __builtin_amdgcn_readfirstlane(hl);
        const int fillR = (c < CH - 1) ? __builtin_amdgcn_readlane(w[c < CH - 1 ? c + 1 : c], 0) : __builtin_amdgcn_readfirstlane(hr);
        const int wl = lane_shr1(w[c], fillL);
        const int wr = __builtin_amdgcn_update_dpp(fillR, w[c], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
        const bool valid = p < M;
        dI[c] = w[c] & 0x7fffffff; yI[c] = (unsigned)w[c] >> 31; dN[c] = wr & 0x7fffffff;
        const bool up = dI[c] <= dN[c], down = dI[c] >= dN[c];
        const bool sameL = (p > 0) && (((unsigned)wl >> 31) == yI[c]), sameR = (p + 1 < M) && (((unsigned)wr >> 31) == yI[c]);
        const bool skip = !fin && ((up && sameL) || (down && sameR));
        rep[c] = valid && !skip;
        pendUp[c] = rep[c] && !fin && up;                      // the scans that go beyond their first step
        pendDn[c] = rep[c] && !fin && down;
        if (!YCIN && g.ycols) {                             // this site's sorted bit column (what pack3 encodes)
            const unsigned long long mk = __ballot(valid && yI[c]);
            const int wd = wv * CH + c;
            if constexpr (P16) { if (lane == ycslot + c) ycacc = mk; }
            else
            if (lane == 0 && wd < g.wpc64) (g.ycols + (size_t)site * g.wpc64)[wd] = mk;
        }
        mPendUp |= __ballot(pendUp[c]); mPendDn |= __ballot(pendDn[c]);
    }
#ifdef PBWTAMD_MEASURE
    if (g.dbg >= 2) mPendUp = mPendDn = 0;                   // measurement (results WRONG): no scans beyond the first step
#endif
    if (mPendUp) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            for (unsigned long long pend = __ballot(pendUp[c]); pend; pend &= pend - 1) {
                WALKSTAT(12, 1);
                const int src = __ffsll((long long)pend) - 1;
                const int thr = __builtin_amdgcn_readlane(dI[c], src);
                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)yI[c], src);
                // candidates q < i, nearest first: the scan stops at q when d[q+1] > thr (tested first), skips i when y[q] == b
                int decided = 0;                            // 1 = not reported (same allele met), 2 = the block ended first
#pragma unroll
                for (int cc = CH - 1; cc >= 0; --cc) {
                    if (cc > c || decided) continue;
                    const int q = wbase + 64 * cc + lane;
                    unsigned long long ms = __ballot(dN[cc] > thr), my = __ballot(q < M && yI[cc] == b);
                    if (cc == c) { const unsigned long long below = (src == 0) ? 0ULL : (~0ULL >> (64 - src)); ms &= below; my &= below; }
                    const unsigned long long any = ms | my;
                    if (any) decided = ((ms >> (63 - __clzll(any))) & 1ULL) ? 2 : 1;
                }
#ifdef PBWTAMD_MEASURE
                if (!decided && g.dbg == 3) decided = 2;    // measurement (results WRONG): no walks through memory
#endif
                int from = wbase - 1;
                if constexpr (P16) { if (!decided && kp1 - thr <= g.clip) decided = walk_lds(from, -1, thr, b); }
                if (!decided) decided = coop_walk4<PACKED, P16>(a, d, from, -1, thr, b, M, p16, kp1) ? 1 : 2;
                if (decided == 1 && lane == src) { rep[c] = false; pendDn[c] = false; }
            }
        }
    }
    if (mPendDn) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            for (unsigned long long pend = __ballot(pendDn[c]); pend; pend &= pend - 1) {
                WALKSTAT(13, 1);
                const int src = __ffsll((long long)pend) - 1;
                const int thr = __builtin_amdgcn_readlane(dN[c], src);
                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)yI[c], src);
                // candidates q > i, nearest first: the scan stops at q when d[q] > thr (d[M] is the sentinel), skips i when y[q] == b
                int decided = 0;
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    if (cc < c || decided) continue;
                    const int q = wbase + 64 * cc + lane;
                    unsigned long long ms = __ballot(q <= M && dI[cc] > thr), my = __ballot(q < M && yI[cc] == b);
                    if (cc == c) { const unsigned long long above = (src == 63) ? 0ULL : (~0ULL << (src + 1)); ms &= above; my &= above; }
                    const unsigned long long any = ms | my;
                    if (any) decided = ((ms >> (__ffsll((long long)any) - 1)) & 1ULL) ? 2 : 1;
                }
#ifdef PBWTAMD_MEASURE
                if (!decided && g.dbg == 3) decided = 2;
#endif
                int from = wbase + 64 * CH;
                if constexpr (P16) { if (!decided && kp1 - thr <= g.clip) decided = walk_lds(from, +1, thr, b); }
                if (!decided) decided = coop_walk4<PACKED, P16>(a, d, from, +1, thr, b, M, p16, kp1) ? 1 : 2;
                if (decided == 1 && lane == src) rep[c] = false;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (rep[c]) {
            const int len = k - min(dI[c], dN[c]);          // (d[i] < d[i+1]) ? k - d[i] : k - d[i+1]   (pbwtMatch.c:131)
            if (len < 0 || len >= g.histlen) atomicExch(g.err, 1);
            else if (len < HIST_LBINS) atomicAdd(&s_hist[P16 ? (len >> 1) : len], P16 ? (1u << (16 * (len & 1))) : 1u);
            else atomicAdd(g.hist + len, 1ULL);
        }
    }
    return true;
    };
    Grp ga, gb;
    if constexpr (YCIN) {
        // 64 words of the site's allele column per wave (lane = word) -> the groups that have to be looked at
        const int wave_lin = blockIdx.x * WAVES + wave_id(), w64 = wave_lin * 64 + lane, g0 = wave_lin * 16;
        const unsigned long long *yc = g.ycols + (size_t)site * g.wpc64;
        const int nw = (M + 63) / 64;
        const unsigned long long yw = (w64 < nw) ? __builtin_nontemporal_load(yc + w64) : 0ULL;
        const unsigned long long yl = (wave_lin > 0) ? __builtin_nontemporal_load(yc + wave_lin * 64 - 1) : 0ULL;          // wave-uniform: the word before / after the wave's 64
        const unsigned long long yr = (wave_lin * 64 + 64 < nw) ? __builtin_nontemporal_load(yc + wave_lin * 64 + 64) : 0ULL;
        const unsigned long long z0 = __ballot(yw == 0ULL), z1 = __ballot(yw == ~0ULL);             // words all 0 / all 1
        const unsigned long long top = __ballot((long long)yw < 0), low = __ballot((yw & 1ULL) != 0);
        const unsigned long long tl = (top << 1) | ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(yl >> 63)) & 1ULL);       // bit w: top bit of word w - 1
        const unsigned long long lr = (low >> 1) | ((unsigned long long)(__builtin_amdgcn_readfirstlane((int)(yr & 1ULL))) << 63);     // bit w: low bit of word w + 1
        unsigned nu = 0;                                    // bit gq: group g0 + gq has to be loaded
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) {
            const unsigned f0 = (unsigned)(z0 >> (4 * gq)) & 15u, f1 = (unsigned)(z1 >> (4 * gq)) & 15u;
            const bool u0 = f0 == 15u && !((tl >> (4 * gq)) & 1ULL) && !((lr >> (4 * gq + 3)) & 1ULL);
            const bool u1 = f1 == 15u && ((tl >> (4 * gq)) & 1ULL) && ((lr >> (4 * gq + 3)) & 1ULL);
            const int wb = (g0 + gq) * (64 * CH);
            const bool inside = wb > 0 && wb + 64 * CH < M;  // the groups holding position 0 or M always take the general path
            if (wb <= M && !((u0 || u1) && inside)) nu |= 1u << gq;
        }
        // the groups to look at, two in flight (any order: the histogram is a sum)
        auto next = [&]() -> int { if (!nu) return -1; const int b = __ffs((int)nu) - 1; nu &= nu - 1; return g0 + b; };
        int ca = next(); if (ca >= 0) request(ga, ca);
        int cb = next(); if (cb >= 0) request(gb, cb);
        while (ca >= 0 || cb >= 0) {
            if (ca >= 0) { process(ga.w[0], ga.w[1], ga.w[2], ga.w[3], ga.hl, ga.hr, ca); ca = next(); if (ca >= 0) request(ga, ca); }
            if (cb >= 0) { process(gb.w[0], gb.w[1], gb.w[2], gb.w[3], gb.hl, gb.hr, cb); cb = next(); if (cb >= 0) request(gb, cb); }
        }
    } else {
    const int wv0 = blockIdx.x * g.iters * WAVES + (P16 ? wave_id_s() * g.iters : wave_id());   // group of iteration it: wv0 + it * WAVES  (P16: wv0 + it, in an SGPR — scalar control flow below)
    if constexpr (P16) {                                    // (the words come from LDS: nothing to request ahead)
#pragma nounroll
        for (int it = 0; it < g.iters; ++it) {
            const int wvq = wv0 + it, rb = wvq * (64 * CH) - r0;       // (inside the workgroup's range by construction)
            if (rb + 64 * CH > rn) break;
            ycslot = it * CH;
            const int *pp = reinterpret_cast<const int *>(s_w + rb) + lane;
            const int hlv = (int)s_w[max(rb - 1, 0)], hrv = (int)s_w[min(rb + 64 * CH, rn - 1)], w0 = pp[0], w1 = pp[64];     // (four LDS reads in flight, branch-free)
            const int hl = (rb > 0) ? hlv : edgeL, hr = (rb + 64 * CH < rn) ? hrv : edgeR;
            if (!process(w0, w1, 0, 0, hl, hr, wvq)) break;
        }
        if (g.ycols && lane < g.iters * CH && wv0 * CH + lane < g.wpc64) (g.ycols + (size_t)site * g.wpc64)[wv0 * CH + lane] = ycacc;     // (groups beyond M leave zeros: padding words)
    } else {
    request(ga, wv0); request(gb, wv0 + WAVES);
    for (int it = 0; it < g.iters; it += 2) {
        if (!process(ga.w[0], ga.w[1], ga.w[2], ga.w[3], ga.hl, ga.hr, wv0 + it * WAVES)) break;
        request(ga, wv0 + (it + 2) * WAVES);
        if (it + 1 >= g.iters) break;
        if (!process(gb.w[0], gb.w[1], gb.w[2], gb.w[3], gb.hl, gb.hr, wv0 + (it + 1) * WAVES)) break;
        request(gb, wv0 + (it + 3) * WAVES);
    }
    }
    }
    __syncthreads();
    unsigned long long *rep = g.hist_rep + (size_t)((blockIdx.x + 7 * blockIdx.y) % HIST_REP) * HIST_LBINS;
    for (int x = threadIdx.x; x < HIST_LBINS; x += BLOCK) { const unsigned v = P16 ? ((s_hist[x >> 1] >> (16 * (x & 1))) & 0xffffu) : s_hist[x]; if (v) atomicAdd(rep + x, (unsigned long long)v); }
}

// matchLongWithin2 (pbwtMatch.c:85-113, -longWithin L) over ring slots: positions are cut into
// blocks wherever d[i] > k-L; every pair ia < ib inside a CLOSED block with different alleles is
// reported with start = max d over (ia, ib].  One thread per ia walks to the end of its block.
// Reference quirks kept: the block still open at position M-1 is never reported (its i0/na/nb live
// across sites and the next site's d[0] closes it with an empty loop), and at the final site k == N
// the alleles are the stale column N-1 (`Ystale` = tags of the previous slot, by position).
// MODE 0 counts per block, MODE 1 emits at the scanned offsets.
struct LongArgs {
    const int *A; const int *D; size_t strideA, strideD;
    const int *Ystale;                   // tagged a of state N-1 (only used for final_site)
    int M, kbase, final_site, L;
    unsigned long long *blockCount; int4 *recs;
};
template <int MODE>
__global__ __launch_bounds__(BLOCK) void sweep_long_kernel(LongArgs g) {
    __shared__ unsigned long long s_w[WAVES];
    const int site = blockIdx.y, k = g.kbase + site;
    const int *a = g.A + (size_t)site * g.strideA;
    const int *d = g.D + (size_t)site * g.strideD;
    const int *ysrc = (site == g.final_site) ? g.Ystale : a;
    const int ia = blockIdx.x * BLOCK + threadIdx.x;
    const int M = g.M, thr = k - g.L;
    unsigned long long cnt = 0;
    int end = 0;
    unsigned ya = 0;
    if (ia < M) {
        ya = (unsigned)ysrc[ia] >> 31;
        int ib = ia + 1;
        while (ib < M && d[ib] <= thr) { if (((unsigned)ysrc[ib] >> 31) != ya) ++cnt; ++ib; }
        end = ib;
        if (ib >= M) cnt = 0;                             // block never closed at this site: not reported
    }
    unsigned long long inc = cnt;
    for (int o = 1; o < 64; o <<= 1) { unsigned long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
    if (lane_id() == 63) s_w[wave_id()] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
    const size_t bidx = (size_t)site * gridDim.x + blockIdx.x;
    if (MODE == 0) { if (threadIdx.x == 0) g.blockCount[bidx] = tot; return; }
    if (cnt) {
        int4 *out = g.recs + g.blockCount[bidx] + pre + (inc - cnt);
        const int ai = a[ia] & AMASK;
        int dmin = 0;
        for (int ib = ia + 1; ib < end; ++ib) {
            dmin = max(dmin, d[ib]);
            if (((unsigned)ysrc[ib] >> 31) != ya) *out++ = make_int4(ai, a[ib] & AMASK, dmin, k);
        }
    }
}

#ifdef PBWTAMD_MEASURE   // the residual sweep behind the fused fill: measurement builds only (pbwt_k_fillseq.h, FUSE)
// matchMaximalWithin, histogram sink, for the positions skel_fillseq_kernel<.., FUSE> left undecided (pbwtMatch.c:115-131): a lane per 32-bit
// flag word, the word's positions one after the other — own word and three neighbours from HBM, a few scalar steps of the two scans, long
// walks finished wave-cooperatively (coop_walk) exactly as sweep_within_kernel<2> does for every position.  Flag words are cleared as
// they are read (the next batch finds the buffer zeroed).  grid (ceil(words / 256), sites).
struct SweepResidArgs { const int *D; size_t strideD; unsigned *flags; size_t strideF; int M, kbase; unsigned long long *hist; int histlen; unsigned long long *hist_rep; int *err; };
__global__ __launch_bounds__(BLOCK) void sweep_resid_kernel(SweepResidArgs g) {
    __shared__ unsigned s_hist[HIST_LBINS];
    const int site = blockIdx.y, k = g.kbase + site, M = g.M, lane = lane_id();
    const int *d = g.D + (size_t)site * g.strideD;
    const int *a = nullptr;
    for (int x = threadIdx.x; x < HIST_LBINS; x += BLOCK) s_hist[x] = 0;
    __syncthreads();
    const int nwords = (M + 31) / 32, wi = blockIdx.x * BLOCK + threadIdx.x;
    unsigned fw = 0;
    if (wi < nwords) { unsigned *fp = g.flags + (size_t)site * g.strideF + wi; fw = *fp; if (fw) *fp = 0u; }
    auto WD = [&](int x) -> int { return __builtin_nontemporal_load(d + x); };
    constexpr int BUDGET = 4;
    while (__ballot(fw != 0)) {
        const bool act = fw != 0;
        const int i = act ? wi * 32 + (__ffs((int)fw) - 1) : M;
        fw &= fw - 1;
        int m = i - 1, n = i + 1, di = 0, dn = 0;
        unsigned yi = 0;
        bool rep = false, needUp = false, needDown = false;
        if (i < M) {
            const int w0 = WD(i), w1 = WD(i + 1);
            di = w0 & 0x7fffffff; yi = (unsigned)w0 >> 31; dn = w1 & 0x7fffffff;
            rep = true;
            if (di <= dn) {                                 // while (d[m+1] <= d[i]) if (y[m--] == y[i]) skip   (pbwtMatch.c:124-126)
                int steps = 0, wcur = w0;
                for (;;) {
                    if ((wcur & 0x7fffffff) > di) break;
                    const int wm = WD(m);                   // m >= 0 here: d[0] is the sentinel, larger than every d[i]
                    if (((unsigned)wm >> 31) == yi) { rep = false; break; }
                    --m; wcur = wm;
                    if (++steps == BUDGET) { needUp = true; break; }
                }
            }
            if (rep && !needUp && di >= dn) {               // while (d[n] <= d[i+1]) if (y[n++] == y[i]) skip    (pbwtMatch.c:127-129)
                int steps = 0, wn = w1;
                for (;;) {
                    if ((wn & 0x7fffffff) > dn) break;
                    if (((unsigned)wn >> 31) == yi) { rep = false; break; }
                    ++n;
                    if (++steps == BUDGET) { needDown = true; break; }
                    wn = WD(n);
                }
            }
        }
        for (unsigned long long pend = __ballot(needUp); pend; pend &= pend - 1) {
            const int src = __ffsll((long long)pend) - 1;
            const int from = __builtin_amdgcn_readlane(m, src), thr = __builtin_amdgcn_readlane(di, src);
            const unsigned yy = (unsigned)__builtin_amdgcn_readlane((int)yi, src);
            int stop = 0;
            const bool skip = coop_walk<true>(a, d, from, -1, thr, yy, false, M, stop);
            if (lane == src) { if (skip) rep = false; else m = stop; }
        }
        if (needUp && rep && di >= dn) {                    // lanes whose upward walk was long still owe the downward scan
            int steps = 0;
            while ((WD(n) & 0x7fffffff) <= dn) {
                if (((unsigned)WD(n) >> 31) == yi) { rep = false; break; }
                ++n;
                if (++steps == BUDGET) { needDown = true; break; }
            }
        }
        for (unsigned long long pend = __ballot(needDown && rep); pend; pend &= pend - 1) {
            const int src = __ffsll((long long)pend) - 1;
            const int from = __builtin_amdgcn_readlane(n, src), thr = __builtin_amdgcn_readlane(dn, src);
            const unsigned yy = (unsigned)__builtin_amdgcn_readlane((int)yi, src);
            int stop = 0;
            const bool skip = coop_walk<true>(a, d, from, +1, thr, yy, false, M, stop);
            if (lane == src) { if (skip) rep = false; else n = stop; }
        }
        if (rep) {
            const int len = k - min(di, dn);                // (d[i] < d[i+1]) ? k - d[i] : k - d[i+1]   (pbwtMatch.c:131)
            if (len < 0 || len >= g.histlen) atomicExch(g.err, 1);
            else if (len < HIST_LBINS) atomicAdd(&s_hist[len], 1u);
            else atomicAdd(g.hist + len, 1ULL);
        }
    }
    __syncthreads();
    unsigned long long *rp = g.hist_rep + (size_t)((blockIdx.x + 7 * blockIdx.y) % HIST_REP) * HIST_LBINS;
    for (int x = threadIdx.x; x < HIST_LBINS; x += BLOCK) { const unsigned v = s_hist[x]; if (v) atomicAdd(rp + x, (unsigned long long)v); }
}
#endif  // PBWTAMD_MEASURE

}  // namespace pbwtk
