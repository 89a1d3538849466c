// pbwt_k_query.h — matchSequencesSweep / Sparse (pbwtMatch.c:363-602): query-side kernels, cursor export, panel transforms.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// matchSequencesSweep (pbwtMatch.c:363-443): Q query haplotypes against the panel.
// Per batch: the panel chain (SORTED, WITH_D) and the query chain (SORTED, A only) fill their ring
// slots; then for the batch's sites
//   qs_unsort  : query alleles back to original query order + each query's rank in the query PBWT
//                order (the reference iterates queries in that order, which fixes the report order)
//   qs_rankdir : zero-prefix directory of the panel column (pbwtCursorCalculateU, pbwtCore.c:510)
//   qs_sweep   : one thread per query walks the batch's sites with its (f, d) state
__global__ __launch_bounds__(BLOCK) void qs_unsort_kernel(const int *AQ, size_t strideAQ, int Mq, unsigned char *xq, int *invq) {
    const int s = blockIdx.y;
    const int *aq = AQ + (size_t)s * strideAQ;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < Mq; j += gridDim.x * BLOCK) {
        const int v = aq[j];
        const int jj = v & AMASK;
        xq[(size_t)s * Mq + jj] = (unsigned char)((unsigned)v >> 31);
        invq[(size_t)s * Mq + jj] = j;
    }
}

// rankdir[s][w] = zeros in positions [0, 64 w) of the panel column; rankdir[s][wpc64] = c
__global__ __launch_bounds__(BLOCK) void qs_rankdir_kernel(const unsigned long long *ycols, int wpc64, int M, int *rankdir) {
    __shared__ int s_w[WAVES];
    __shared__ int s_carry;
    const int s = blockIdx.x;
    const unsigned long long *y = ycols + (size_t)s * wpc64;
    int *rd = rankdir + (size_t)s * (wpc64 + 1);
    const int nw = (M + 63) / 64;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b = 0; b < wpc64; b += BLOCK) {
        const int wd = b + threadIdx.x;
        int z = 0;
        if (wd < nw) z = min(64, M - wd * 64) - __popcll(y[wd]);
        const int inc = wave_iscan_sum(z);
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        int pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
        if (wd < wpc64) rd[wd] = pre + inc - z;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) rd[wpc64] = s_carry;
}

// ---------------------------------------------------------------------------------------------
// matchSequencesSweepSparse (pbwtMatch.c:452-602): the query sweep against the panel cursor AND, at
// site k, against the sparse cursor kk = k % nS (a PBWT of the sites = kk mod nS, stepped with
// pbwtCursorForwardsAD(.., k/nS)).  One thread per query walks the batch's sites carrying (f, d) for
// the dense cursor and for each of the nS sparse cursors; counts -> scan -> emit keeps callback order
// (per site and query rank: dense block, then sparse block).
struct Rec5 { int ai, bi, start, end, sparse; };
struct QsView {                                              // one cursor's states for the sites of a batch
    const int *A; const int *D; size_t strideA, strideD;
    const unsigned long long *ycols; const int *rankdir;    // sorted bit columns, zero-prefix directory [slot][wpc64+1]
    int sbase;                                               // sparse cursors: index of the cursor's first step in this batch
    const int *A0;                                           // a copy of the batch's FIRST a[] row for the emission pass: the next batch's chain, which runs
                                                             // beside it, ends by writing its own first state into that ring slot
    const int2 *bsum; int nblk;                              // per slot and block of 256 positions: {max d (INT_MAX when the block reaches position M), bit 0: holds a 0, bit 1: holds a 1}; null = none
};

// block summaries for the walks of reportAndUpdate (pbwtMatch.c:452-499).  The reference walks position by position (1 ns each
// on a CPU); here a wave tests 256 positions per trip to memory (~1.5 us), and a query whose allele is rare around its match walks
// 10^5..10^6 of them: measured at M = 1 M, Q = 10 k, the slowest of the 10 000 waves of a 512-site batch took 3.4-4.6 ms where the
// mean took 0.43.  With {max d, alleles present} per 256 positions a walk skips 64 blocks per lane-step: 65 536 positions per trip.
// grid (ceil(nblk / 16), sites): a wave takes four consecutive blocks (one 16-byte load per lane and block, all four in flight).
__global__ __launch_bounds__(BLOCK) void qs_blocksum_kernel(const int *D, size_t strideD, const unsigned long long *ycols, int wpc64, int M, int nblk, int2 *bsum) {
    const int s = blockIdx.y, b0 = (blockIdx.x * WAVES + wave_id()) * 4, lane = lane_id();
    if (b0 >= nblk) return;
    const int *d = D + (size_t)s * strideD;                  // slots are 16-byte aligned (strides are multiples of 64 ints)
    const unsigned long long *yc = ycols + (size_t)s * wpc64;
    int4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = (b0 + q) * 256 + 4 * lane;            // reads stay inside the slot's padding (Mpad is a multiple of 4096)
        v[q] = (b0 + q < nblk) ? *reinterpret_cast<const int4 *>(d + i) : make_int4(0, 0, 0, 0);
    }
    unsigned long long word = 0ULL; int valid = 0;
    if (lane < 16) { const int w = b0 * 4 + lane; valid = min(64, M - w * 64); if (valid > 0) word = yc[w]; }
    int fl = 0;
    if (valid > 0) { const unsigned long long mask = (valid == 64) ? ~0ULL : ((1ULL << valid) - 1ULL); fl = ((~word & mask) ? 1 : 0) | ((word & mask) ? 2 : 0); }
    fl |= __shfl_xor(fl, 1); fl |= __shfl_xor(fl, 2);      // lanes 4q .. 4q+3: block b0 + q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = (b0 + q) * 256 + 4 * lane;
        int mx = max(max(i < M ? v[q].x : 0x7fffffff, i + 1 < M ? v[q].y : 0x7fffffff), max(i + 2 < M ? v[q].z : 0x7fffffff, i + 3 < M ? v[q].w : 0x7fffffff));
        mx = wave_max(mx);
        const int f = __builtin_amdgcn_readlane(fl, 4 * q);
        if (lane == 0 && b0 + q < nblk) bsum[(size_t)s * nblk + b0 + q] = make_int2(mx, f);
    }
}
struct QssArgs {
    QsView dense; const QsView *sparse;                      // sparse[nS] in device memory
    int wpc64, nS;
    const unsigned char *xq; const int *invq;                // [site][Mq]
    int Mp, Mq, kbase, nsites;
    const int *f_in; const int *dq_in; int *f_out; int *dq_out;            // [Mq]
    const int *fs_in; const int *ds_in; int *fs_out; int *ds_out;          // [nS][Mq]
    unsigned long long *cnt;                                 // [site][Mq rank][2]: counts / exclusive offsets (dense, sparse)
    Rec5 *recs;
    unsigned long long *tot;                                 // [0] nTot [1] totLen [2] no-match events
    int4 *nm_ev; unsigned *nm_n; unsigned nm_cap;            // the no-match events themselves: {site k, query rank, query jj, x | isSparse << 1}
    int2 *evt;                                               // per slot with reports: {first panel position f, reported start} — what qss_emit_kernel expands
    int q_lo, q_hi;                                          // only the queries q_lo <= jj < q_hi are swept (query sharding across GPUs: pbwtamd_set_query_range)
    unsigned long long *dbg;                                 // measurement (PBWTAMD_QS_DBG): per query {wall-clock ticks (100 MHz) of its wave, events} accumulated over the batches
};

// reportAndUpdate (pbwtMatch.c:452-499) for one query at one site against one cursor state, executed by a whole
// WAVE: every walk of the reference (the scan for an equally long match further down, the alternating widening of
// [iMinus, iPlus]) tests 64 positions per step with ballots.  A single lane walking them one dependent load at a time
// costs ~1 us per position on this machine (measured: 5.5 ms per site at M = 100 k) where the CPU pays ~1 ns.
// All arguments and results are wave-uniform.
template <int MODE>
__device__ __forceinline__ void qss_update(const int *a, const int *d, const unsigned long long *yc, int M, unsigned x, int jj, int k,
                                           int kend, int nS, int isSparse, int &f, int &dq, unsigned long long *cntslot, Rec5 *recs,
                                           unsigned long long &nTot, unsigned long long &totLen, unsigned long long &nomatch,
                                           int rank, int4 *nm_ev, unsigned *nm_n, unsigned nm_cap, int2 *evt, const int2 *bs = nullptr, int nblk = 0) {
    const int lane = lane_id();
#define PY(i) ((unsigned)((yc[(i) >> 6] >> ((i) & 63)) & 1ULL))
    if (PY(f) == x) return;
    const int xbit = x ? 2 : 1;
    // first block >= b0 that may end a downward scan with threshold thr: max d above it, the allele present, or past the panel
    auto coarse_down = [&](int b0, int thr) -> int {
        for (int base = b0;; base += 256) {
            int2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int b = base + 64 * q + lane; v[q] = (b < nblk) ? bs[b] : make_int2(0x7fffffff, 3); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long m = __ballot(v[q].x > thr || (v[q].y & xbit));
                if (m) return base + 64 * q + __ffsll((long long)m) - 1;
            }
        }
    };
    // downward scan from `from` while d <= thr: the first position that either fails the test (or is M) or carries x
    // (256 positions per trip to memory: the four 64-position sub-steps' loads are issued together, then tested in order — a query
    // whose allele is rare around its match walks thousands of positions here, one dependent round trip per step)
    auto scan_down = [&](int from, int thr, bool &found) -> int {
        for (int base = from;;) {
            int dv[4]; unsigned long long yw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = base + 64 * q + lane; dv[q] = (i < M) ? d[i] : 0x7fffffff; yw[q] = (i < M) ? yc[i >> 6] : 0ULL; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = base + 64 * q + lane;
                const bool bound = dv[q] > thr;
                const bool same = !bound && (unsigned)((yw[q] >> (i & 63)) & 1ULL) == x;
                const unsigned long long mb = __ballot(bound), ms = __ballot(same), any = mb | ms;
                if (any) { const int first = __ffsll((long long)any) - 1; found = (ms >> first) & 1ULL; return base + 64 * q + first; }
            }
            base += 256;
            // 256 positions without an end: skip the blocks that cannot hold one (a block flagged for what lies in its part already
            // scanned costs one more fine trip, never a wrong answer)
            if (bs) base = max(base, coarse_down(base >> 8, thr) << 8);
        }
    };
    bool found = false;
    int iPlus = scan_down(f + 1, dq, found);                 // pbwtMatch.c:455-457
    if (found) { f = iPlus; return; }
    const int n = iPlus - f;                                 // these matches end here (pbwtMatch.c:459-461)
    const int dj = isSparse ? nS * dq + k % nS : dq;
    if (MODE == 0) { if (lane == 0) { *cntslot = (unsigned long long)n; if (evt) *evt = make_int2(f, dj); } nTot += n; totLen += (unsigned long long)(k - dj) * n; }
    else {
        Rec5 *o = recs + *cntslot;
        for (int i = f + lane; i < iPlus; i += 64) { Rec5 r; r.ai = jj; r.bi = a[i] & AMASK; r.start = dj; r.end = k; r.sparse = isSparse; o[i - f] = r; }
    }
    int iMinus = f;
    int dPlus = (iPlus < M) ? d[iPlus] : kend;
    int dMinus = d[iMinus];
    for (;;) {                                               // widen [iMinus, iPlus] by the smaller divergence until an x is met (:477-498)
        if (dMinus <= dPlus) {
            // while (d[iMinus] <= dMinus) if (y[--iMinus] == x) hit = iMinus;   d[0] = kend+1 stops it; the LOWEST hit counts
            int hit = -1;
            int skipLo = 0, skipHi = 0;                      // positions [skipLo, skipHi) were passed in whole blocks, their candidates not looked at yet
            for (int base4 = iMinus, go = 1; go;) {
                int dv[4]; unsigned long long yw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int j = base4 - 64 * q - lane; dv[q] = (j >= 0) ? d[j] : 0x7fffffff; yw[q] = (j - 1 >= 0) ? yc[(j - 1) >> 6] : 0ULL; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!go) continue;
                    const int base = base4 - 64 * q, j = base - lane;
                    const unsigned long long mstop = __ballot(dv[q] > dMinus);
                    const int nlive = mstop ? __ffsll((long long)mstop) - 1 : 64;     // lanes 0..nlive-1 passed the test: candidates j-1
                    const bool cand = (lane < nlive) && (j - 1 >= 0) && (unsigned)((yw[q] >> ((j - 1) & 63)) & 1ULL) == x;
                    const unsigned long long mc = __ballot(cand);
                    if (mc) { hit = base - (63 - __clzll(mc)) - 1; skipHi = skipLo = 0; }   // highest lane = lowest index; lower than anything skipped before
                    if (mstop) { iMinus = base - nlive; go = 0; }
                }
                if (!go) break;
                base4 -= 256;                                // 256 positions passed, the next one to test is base4
                if (bs && base4 > 0) {
                    // blocks in which every d <= dMinus are passed without a stop.  Going down from the block that holds base4 (its part above
                    // base4 was passed or lies above the walk's start: at worst it makes the block look like a stop and nothing is skipped), the first
                    // block with a larger d — block 0 has one, the sentinel d[0] — is where the fine walk goes on, at its last position
                    const int bt = base4 >> 8;
                    int bstop = -1;
                    for (int bb = bt; bstop < 0; bb -= 256) {
                        int mxv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const int b = bb - 64 * q - lane; mxv[q] = (b >= 0) ? bs[b].x : 0x7fffffff; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (bstop >= 0) continue;
                            const unsigned long long m = __ballot(mxv[q] > dMinus);
                            if (m) bstop = bb - 64 * q - (__ffsll((long long)m) - 1);
                        }
                    }
                    if (bstop < 0) bstop = 0;
                    const int nb4 = (bstop + 1) * 256 - 1;   // last position of the stopping block
                    if (nb4 < base4) {
                        // every j in nb4+1 .. base4 passes; their candidates j - 1 are the positions [nb4, base4 - 1].  A candidate lower than all of
                        // them may still turn up further down: remember the range, look into it only if that does not happen
                        if (skipHi == skipLo) skipHi = base4;
                        skipLo = nb4;
                        base4 = nb4;
                    }
                }
            }
            if (skipHi > skipLo && (hit < 0 || hit >= skipHi)) {
                // the lowest position in [skipLo, skipHi) carrying x, if any (the skipped stretch lies below every earlier hit)
                int p = skipLo;
                while (p < skipHi) {
                    unsigned long long yw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int i = p + 64 * q + lane; yw[q] = (i < skipHi) ? yc[i >> 6] : 0ULL; }
                    int got = -1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (got >= 0) continue;
                        const int i = p + 64 * q + lane;
                        const unsigned long long m = __ballot(i < skipHi && (unsigned)((yw[q] >> (i & 63)) & 1ULL) == x);
                        if (m) got = p + 64 * q + __ffsll((long long)m) - 1;
                    }
                    if (got >= 0) { hit = got; break; }
                    p += 256;
                    if (p < skipHi) {                        // blocks without the allele: skip them
                        int b0 = p >> 8, bfound = -1;
                        for (int bb = b0; bfound < 0; bb += 256) {
                            int fl[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { const int b = bb + 64 * q + lane; fl[q] = (b < nblk && (b << 8) < skipHi) ? bs[b].y : 3; }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (bfound >= 0) continue;
                                const unsigned long long m = __ballot((fl[q] & xbit) != 0);
                                if (m) bfound = bb + 64 * q + __ffsll((long long)m) - 1;
                            }
                        }
                        p = max(p, bfound << 8);
                    }
                }
            }
            if (hit >= 0) { f = hit; dq = dMinus; return; }
            dMinus = d[iMinus];
        } else {
            bool got = false;
            iPlus = scan_down(iPlus, dPlus, got);
            if (got) { f = iPlus; dq = dPlus; return; }
            dPlus = (iPlus < M) ? d[iPlus] : kend;
            if (!iMinus && iPlus == M) {                     // "no match to query jj value x at site k" (pbwtMatch.c:405-410)
                ++nomatch; dq = 1 + kend;
                if (MODE == 0 && lane == 0 && nm_ev) { const unsigned at = atomicAdd(nm_n, 1u); if (at < nm_cap) nm_ev[at] = make_int4(k, rank, jj, (int)x | (isSparse << 1)); }
                return;
            }
        }
    }
#undef PY
}

__device__ __forceinline__ int qss_lfmap(const unsigned long long *yc, const int *rd, int wpc64, int M, unsigned x, int f) {
    const unsigned long long wdv = yc[f >> 6];               // pbwtCursorMap (pbwt.h:130-131) with the f == M trap (pbwtMatch.c:552,561)
    const int uf = rd[f >> 6] + ((f & 63) - __popcll(wdv & ((1ULL << (f & 63)) - 1ULL)));
    const int c = rd[wpc64];
    f = x ? c + f - uf : uf;
    return (f == M) ? 0 : f;
}

// one WAVE per query
// qpw > 1: a wave takes qpw queries one after the other (query = wave + i * waves of the launch): a quarter of the waves resident for the
// whole batch leaves the chain's dependent launches room on every CU, and the sweep has the time
template <int MODE>
__global__ __launch_bounds__(BLOCK) void qss_sweep_kernel(QssArgs g) {
    const int lane = lane_id(), M = g.Mp, nS = g.nS;
    const int wave0 = blockIdx.x * WAVES + wave_id(), nwaves = gridDim.x * WAVES;
    for (int jj = wave0; jj < g.Mq; jj += nwaves) {
    if (jj < g.q_lo || jj >= g.q_hi) continue;               // another rank's query: its count slots stay zero
    int f = g.f_in[jj], dq = g.dq_in[jj];
    unsigned long long nTot = 0, totLen = 0, nomatch = 0;
    const unsigned long long t_in = g.dbg ? wall_clock64() : 0ULL; unsigned nev = 0;
    // the sparse (f, d) pairs live in global memory (nS is a run-time value): working copy in the out arrays
    if (MODE == 0 && lane == 0) for (int kk = 0; kk < nS; ++kk) { g.fs_out[(size_t)kk * g.Mq + jj] = g.fs_in[(size_t)kk * g.Mq + jj]; g.ds_out[(size_t)kk * g.Mq + jj] = g.ds_in[(size_t)kk * g.Mq + jj]; }
    int fsl = 0, dsl = 0;
    unsigned xpre = 0; int ipre = 0;                         // this query's allele and PBWT rank at 64 sites: lane l holds site s0 + l
    for (int s = 0; s < g.nsites; ++s) {
        const int k = g.kbase + s;
        if ((s & 63) == 0) {
            const int sl = s + lane;
            xpre = (sl < g.nsites) ? g.xq[(size_t)sl * g.Mq + jj] : 0u;
            ipre = (sl < g.nsites) ? g.invq[(size_t)sl * g.Mq + jj] : 0;
        }
        const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)xpre, s & 63);
        const int qrank = __builtin_amdgcn_readlane(ipre, s & 63);
        const size_t slot = ((size_t)s * g.Mq + qrank) * 2;
        {
            const int *a = g.dense.A + (size_t)s * g.dense.strideA, *d = g.dense.D + (size_t)s * g.dense.strideD;
            const unsigned long long *yc = g.dense.ycols + (size_t)s * g.wpc64;
            const int *rd = g.dense.rankdir + (size_t)s * (g.wpc64 + 1);
            // the common case (the match extends) is ONE memory round trip per site: the column word, its rank directory
            // entry and the zero count depend on f only and are requested together
            const unsigned long long w0 = yc[f >> 6];
            const int r0 = rd[f >> 6], c0 = rd[g.wpc64];
            if ((unsigned)((w0 >> (f & 63)) & 1ULL) == x) {
                const int uf = r0 + ((f & 63) - __popcll(w0 & ((1ULL << (f & 63)) - 1ULL)));
                f = x ? c0 + f - uf : uf;
                if (f == M) f = 0;
            } else {
                ++nev;
                qss_update<MODE>(a, d, yc, M, x, jj, k, k, nS, 0, f, dq, g.cnt + slot, g.recs, nTot, totLen, nomatch, qrank, g.nm_ev, g.nm_n, g.nm_cap, g.evt ? g.evt + slot : nullptr,
                                 g.dense.bsum ? g.dense.bsum + (size_t)s * g.dense.nblk : nullptr, g.dense.nblk);
                f = qss_lfmap(yc, rd, g.wpc64, M, x, f);
            }
        }
        if (nS > 1) {
            const int kk = k % nS;
            const QsView v = g.sparse[kk];
            const int t = k / nS - v.sbase;                 // this cursor's slot in its ring
            const int *a = v.A + (size_t)t * v.strideA, *d = v.D + (size_t)t * v.strideD;
            const unsigned long long *yc = v.ycols + (size_t)t * g.wpc64;
            // MODE 1 replays the same walk from the batch's input state (scratch half of the out arrays)
            const size_t ix = (size_t)kk * g.Mq + jj, sx = (size_t)(nS + kk) * g.Mq + jj;
            if (MODE == 0) { fsl = g.fs_out[ix]; dsl = g.ds_out[ix]; }
            else if (s < nS) { fsl = g.fs_in[ix]; dsl = g.ds_in[ix]; }
            else { fsl = g.fs_out[sx]; dsl = g.ds_out[sx]; }
            qss_update<MODE>(a, d, yc, M, x, jj, k, k / nS, nS, 1, fsl, dsl, g.cnt + slot + 1, g.recs, nTot, totLen, nomatch, qrank, g.nm_ev, g.nm_n, g.nm_cap, g.evt ? g.evt + slot + 1 : nullptr,
                             v.bsum ? v.bsum + (size_t)t * v.nblk : nullptr, v.nblk);
            fsl = qss_lfmap(yc, v.rankdir + (size_t)t * (g.wpc64 + 1), g.wpc64, M, x, fsl);
            if (lane == 0) {
                if (MODE == 0) { g.fs_out[ix] = fsl; g.ds_out[ix] = dsl; }
                else { g.fs_out[sx] = fsl; g.ds_out[sx] = dsl; }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();                          // the wave reads its own lane-0 store back at its next visit of this cursor
        }
    }
    if (MODE == 0 && lane == 0) {
        g.f_out[jj] = f; g.dq_out[jj] = dq;
        if (nTot) { atomicAdd(g.tot, nTot); atomicAdd(g.tot + 1, totLen); }
        if (nomatch) atomicAdd(g.tot + 2, nomatch);
        if (g.dbg) {
            const unsigned long long dt = wall_clock64() - t_in;
            g.dbg[2 * (size_t)jj] += dt; g.dbg[2 * (size_t)jj + 1] += nev;
            atomicMax(g.dbg + 2 * (size_t)g.Mq + (size_t)(g.kbase / max(g.nsites, 1)) % 64, dt);           // slowest wave of the batch
            atomicMax(g.dbg + 2 * (size_t)g.Mq + 64 + (size_t)(g.kbase / max(g.nsites, 1)) % 64, (unsigned long long)nev);
        }
    }
    }
}

// records of a batch from the counting pass's event descriptors: slot (site s, query rank r, dense / sparse) with n reports
// -> (query AQ[s][r], a[f + i], start, k, isSparse) for i < n, at the slot's scanned offset.  Replaces a second run of the whole
// sweep in emit mode (the walks are done once).  A wave takes 64 consecutive slots; the non-empty ones are expanded cooperatively.
struct QssEmitArgs {
    const unsigned long long *off; const unsigned long long *total;   // exclusive offsets per slot (scan of the counts), their total
    const int2 *evt; size_t nslots;
    QsView dense; const QsView *sparse; int nS;
    const int *AQ; size_t strideAQ; const int *AQ0;              // query cursor: position r of site s holds the query index (AQ0: copy of row 0, see QsView::A0)
    int Mq, kbase;
    Rec5 *recs;
    // lazy ids (dense cursor): the batch's fill wrote d only (skel_fill_kernel<., 2>); a[] exists at the skeleton slots 0, 8, 16, ... and at
    // slot 0 of the other ring (Anext = the state after the batch's last site).  The id at position p of slot s = 8b + j is the id at
    // LF^(8-j)(p) of slot 8(b+1): <= 7 steps of pbwtCursorMap (pbwt.h:130-131) through the batch's own columns and rank directories
    int lazy, nsites, wpc64; const int *Anext;
    int emit_rank;                                           // query sharding: the query's rank r in the query panel's order at the site goes into sparse >> 1 (the merge key)
};
__global__ __launch_bounds__(BLOCK) void qss_emit_kernel(QssEmitArgs g) {
    const size_t base = ((size_t)blockIdx.x * WAVES + wave_id()) * 64;
    const int lane = lane_id();
    if (base >= g.nslots) return;
    const size_t slot = base + lane;
    unsigned long long off = 0, nxt = 0;
    if (slot < g.nslots) { off = g.off[slot]; nxt = (slot + 1 < g.nslots) ? g.off[slot + 1] : *g.total; }
    const int n = (int)(nxt - off);
    for (unsigned long long pend = __ballot(n > 0); pend; pend &= pend - 1) {
        const int src = __ffsll((long long)pend) - 1;
        const size_t sl = base + src;
        const int cntN = __builtin_amdgcn_readlane(n, src);
        const unsigned long long o0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(off >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)off, src);
        const int2 ev = g.evt[sl];
        const int sparse = (int)(sl & 1), r = (int)((sl >> 1) % (size_t)g.Mq), s = (int)((sl >> 1) / (size_t)g.Mq);
        const int k = g.kbase + s;
        const int jj = (s ? g.AQ[(size_t)s * g.strideAQ + r] : g.AQ0[r]) & AMASK;
        const int *a;
        if (sparse) { const QsView v = g.sparse[k % g.nS]; const int t = k / g.nS - v.sbase; a = t ? v.A + (size_t)t * v.strideA : v.A0; }
        else a = s ? g.dense.A + (size_t)s * g.dense.strideA : g.dense.A0;
        int s8 = s;                                          // the slot the ids are read from
        if (!sparse && g.lazy && (s & 7)) { s8 = (s | 7) + 1; a = (s8 < g.nsites) ? g.dense.A + (size_t)s8 * g.dense.strideA : g.Anext; }
        for (int i = lane; i < cntN; i += 64) {
            int p = ev.x + i;
            for (int t = s; t < s8; ++t) {
                const unsigned long long wdv = g.dense.ycols[(size_t)t * g.wpc64 + (p >> 6)];
                const int *rd = g.dense.rankdir + (size_t)t * (g.wpc64 + 1);
                const int up = rd[p >> 6] + ((p & 63) - __popcll(wdv & ((1ULL << (p & 63)) - 1ULL)));
                p = ((wdv >> (p & 63)) & 1ULL) ? rd[g.wpc64] + p - up : up;
            }
            Rec5 rr; rr.ai = jj; rr.bi = a[p] & AMASK; rr.start = ev.y; rr.end = k; rr.sparse = sparse | (g.emit_rank ? (r << 1) : 0); g.recs[o0 + i] = rr;
        }
    }
}

// matches still running at the end of the panel for one cursor (pbwtMatch.c:577-594), in final query
// order; sparse cursor kk: start nS*d + kk, totLen with the cursor's own d (as the reference).  One wave per query.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void qss_tail_kernel(const int *A, const int *D, const int *AQ, int Mp, int Mq, int N, int nS, int kk, int isSparse,
                                                        const int *f, const int *dq, unsigned long long *cnt, Rec5 *recs, unsigned long long *tot,
                                                        int q_lo, int q_hi, int emit_rank) {
    const int j = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (j >= Mq) return;
    const int jj = AQ[j] & AMASK;
    if (jj < q_lo || jj >= q_hi) { if (MODE == 0 && lane == 0) cnt[j] = 0; return; }   // another rank's query
    const int f0 = f[jj], d0 = dq[jj];
    int i = f0 + 1;                                          // for (i = f; ++i < M && d[i] <= dq; )
    for (;; i += 64) {
        const int p = i + lane;
        const unsigned long long mb = __ballot((p >= Mp) || (D[p] > d0));
        if (mb) { i += __ffsll((long long)mb) - 1; break; }
    }
    const int n = i - f0;
    const int dj = isSparse ? nS * d0 + kk : d0;
    if (MODE == 0) { if (lane == 0) { cnt[j] = (unsigned long long)n; atomicAdd(tot, (unsigned long long)n); atomicAdd(tot + 1, (unsigned long long)(N - d0) * n); } }
    else { Rec5 *o = recs + cnt[j]; for (int q = f0 + lane; q < i; q += 64) { Rec5 r; r.ai = jj; r.bi = A[q] & AMASK; r.start = dj; r.end = N; r.sparse = isSparse | (emit_rank ? (j << 1) : 0); o[q - f0] = r; } }
}

// PbwtCursor view of one sorted bit column (pbwt.h:78-83): y[i] as bytes, u[i] = zeros in y[0..i) for i = 0..M
// (pbwtCursorCalculateU, pbwtCore.c:510-519) from the column's zero-prefix directory
__global__ void cursor_y_u_kernel(const unsigned long long *yc, const int *rd, int M, unsigned char *y, int *u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    const int wd = i >> 6, bo = i & 63;
    const unsigned long long w = (i < M || bo) ? yc[wd] : 0ULL;
    if (i < M) y[i] = (unsigned char)((w >> bo) & 1ULL);
    u[i] = (i < M || bo) ? rd[wd] + (bo - __popcll(w & ((1ULL << bo) - 1ULL))) : rd[wd];
}

// panel transforms (pbwtBuildReverse pbwtCore.c:151-191, pbwtSubSample pbwtSample.c:59-93, pbwtSubRange pbwtCore.c:111-148,
// pbwtSelectSites pbwtCore.c:623-682) are all "x[a[j]] = y[j]; y'[j] = x[a'[j]]" loops: the first half is
// unsort_alleles_kernel (alleles of a batch of sites back in original haplotype order), this is the gather half — the
// bit column of output site inv[s] = the selected haplotypes of input site s, in the new panel's haplotype order.
// grid (ceil(wpc64_out / WAVES), sites of the batch); one wave builds one 64-haplotype word with a ballot.
__global__ __launch_bounds__(BLOCK) void regather_kernel(const unsigned char *alleles, int M_in, const int *site_to_out, const int *hap_select,
                                                        int M_out, unsigned long long *cols_out, int wpc64_out) {
    const int s = blockIdx.y, j = site_to_out[s];
    if (j < 0) return;                                       // site dropped
    const unsigned char *x = alleles + (size_t)s * M_in;
    for (int wd = blockIdx.x * WAVES + wave_id(); wd < wpc64_out; wd += gridDim.x * WAVES) {
        const int h = wd * 64 + lane_id();
        const bool one = (h < M_out) && x[hap_select ? hap_select[h] : h] != 0;
        const unsigned long long mk = __ballot(one);
        if (lane_id() == 0) cols_out[(size_t)j * wpc64_out + wd] = mk;
    }
}

// bytes (0/1 per haplotype, original order) -> bit column words; grid (words/4, sites)
__global__ __launch_bounds__(BLOCK) void bytes_to_bits_kernel(const unsigned char *in, int M, unsigned long long *out, int wpc64) {
    const int s = blockIdx.y;
    for (int wd = blockIdx.x * WAVES + wave_id(); wd < wpc64; wd += gridDim.x * WAVES) {
        const int i = wd * 64 + lane_id();
        const unsigned long long mk = __ballot(i < M && in[(size_t)s * M + i] != 0);
        if (lane_id() == 0) out[(size_t)s * wpc64 + wd] = mk;
    }
}

}  // namespace pbwtk
