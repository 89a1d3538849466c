// pbwt_k_fill.h — the fill, table form (skel_fill_kernel): the seven states between two skeleton states out of the round's tables folded down bit by bit.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// FILL, one launch per batch: the seven states between two skeleton states, for every 8-site block b
// and every tile w of the block's input state (grid (W, blocks)).  State 8b+j is the stable sort of
// state 8b by the low j bits of the same 8-bit keys, so everything the rank kernel derived for j = 8
// folds down: counts / last positions per chunk, keys before the tile, carries and totals of a j-bit
// key are sums / maxima / minima over the 8-bit keys sharing its low bits (a later last occurrence has
// the smaller suffix maximum, hence min over the carries).  Reads a, d, keys once, writes 7 x (a, d).
struct SkFillArgs {
    int *A; int *D; size_t strideA, strideD;               // ring base (slot 0 of the batch)
    const unsigned char *keys; size_t strideK;              // keys of state 8b at keys + b*strideK
    const int2 *scan; size_t strideS;                       // per block: scan[W][256] {before, carry}, then total[256] (strideS in int2 units)
    int M, W, kbase;
#ifdef PBWTAMD_MEASURE
    int dbg_nowrite;                                        // measurement builds only (results WRONG): no stores
#endif
    int pack_y;                                             // write d | y << 31 only (no a): for consumers that need (d, y) but not the haplotype ids
    int xcd;                                                // XCD-contiguous (round, tile) pairs (xcd_tile)
    int pair, W2;                                           // pair rows: scan[W2][256], total, then the first halves' rows tbl0[W2][256] per round
    int aggx_off, aggx_tpw;                                 // > 0: the rows are local to scan workgroups of aggx_tpw rows; their exclusive aggregates stand aggx_off int2 into the round's block (SkArgs::aggx)
};

// PACKY 1: the consumers need (d, y) of every site but not the haplotype ids — a[] is neither read nor written, slots hold d | y << 31
// PACKY 2: d only, plain (the query sweep: y comes from the decoded columns, the ids of the few reported positions are recovered from the
//          next skeleton state by qss_emit_kernel) — a[] neither read nor written, the skeleton slots left as they are
template <int EPT, int PACKY>
__global__ __launch_bounds__(BLOCK) void skel_fill_kernel(SkFillArgs g) {
    constexpr int T = BLOCK * EPT, NC = EPT * WAVES;
    // Range maxima of d_k through a RADIX-4 sparse table: level e holds max d over (i - 4^e, i], windows 1, 4, 16, 64 (, 256): a
    // range of len positions is covered by <= 4 windows of the largest level with 4^e <= len (a radix-2 table answers with 2 reads
    // but costs 18 KB at T = 512).  This kernel is occupancy-bound — measured: 2 instead of 4 workgroups per CU takes 1.73x
    // as long — so LDS is what counts.  The per-chunk tables of the 8-bit keys are dead after the first fold step and the
    // sparse-table levels >= 2 are born after it: they share storage.  26 KB at T = 512: 6 workgroups per CU (was 40 KB, 4).
    constexpr int NL4 = (EPT == 1) ? 4 : 5;
    // heap layout: level j (keys of j bits) lives at [2^j, 2^(j+1)); levels 1..7 in s_rawH / s_lastH, the rank kernel's level 8 in s_raw8 / s_last8
    __shared__ short s_rawH[NC][SKK], s_lastH[NC][SKK];      // per chunk: count / last local position (-1) -> base / previous position (exclusive over the chunks)
    constexpr int UBYTES = (2 * NC * SKK * 2 > (NL4 - 2) * T * 4) ? 2 * NC * SKK * 2 : (NL4 - 2) * T * 4;
    // ONE array: sparse levels 0, 1, then the shared storage — level lv starts at lv * T words whatever lv is (no select per query)
    __shared__ __attribute__((aligned(16))) unsigned char s_tb[2 * T * 4 + UBYTES];
    unsigned char *const s_u = s_tb + 2 * T * 4;
    short (*const s_raw8)[SKK] = reinterpret_cast<short (*)[SKK]>(s_u);                      // until fold step 1
    short (*const s_last8)[SKK] = reinterpret_cast<short (*)[SKK]>(s_u + NC * SKK * 2);
    int (*const s_tbl01)[T] = reinterpret_cast<int (*)[T]>(s_tb);                            // sparse levels 0, 1; levels 2 .. NL4-1 (from step 2 on) follow in s_u
    auto TBL = [&](int lv) -> int * { return reinterpret_cast<int *>(s_tb) + lv * T; };
    constexpr int EFLAG = 0x40000000;                        // s_cH[h] after level_scan: carry | EFLAG (max with the range maximum) or the final value
    __shared__ int s_bH[2 * SKK], s_cH[2 * SKK], s_tH[2 * SKK];
    int *const s_GH = &s_bH[SKK];                            // the level-8 halves are dead once level 7 is folded
    // STAGE (no ids to move, T <= 512): every sub-step's outputs pass through LDS in DESTINATION order, so that a wave's store covers a few
    // runs of consecutive addresses instead of 64 scattered words (at the seventh sub-step a tile feeds 128 runs of ~4 positions).  No
    // LDS is added: values in s_tH (dead after the level scans), key bytes and the tile's own bucket totals in the dead half of s_cH.
    constexpr bool STAGE = (PACKY >= 1) && (EPT <= 2);
    short *const s_loc = reinterpret_cast<short *>(&s_cH[SKK]);                              // [256] this tile's total per heap entry
    unsigned char *const s_kb = reinterpret_cast<unsigned char *>(&s_cH[SKK]) + 2 * SKK;     // [T] key bytes, destination order
    int *const s_stage = s_tH;                                                               // [T] values, destination order
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    int w = blockIdx.x, b = blockIdx.y;
    if (g.xcd) { const int lg = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y); b = lg / g.W; w = lg - b * g.W; }
    const int S = w * T, k = g.kbase + 8 * b;
    const int *a_in = g.A + (size_t)(8 * b) * g.strideA;
    int *d_in = g.D + (size_t)(8 * b) * g.strideD;
    const unsigned char *keys = g.keys + (size_t)b * g.strideK;
    const int2 *sv = g.scan + (size_t)b * g.strideS;
    int av[EPT], key[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int l = r * BLOCK + t, i = S + l;
        const bool valid = i < g.M;
        av[r] = PACKY ? 0 : (a_in[i] & AMASK); key[r] = valid ? (int)keys[i] : -1;
        const int dv = valid ? d_in[i] : 0;
        s_tbl01[0][l] = dv;
        if (PACKY == 1 && valid) d_in[i] = dv | (int)(((unsigned)key[r] & 1u) << 31);   // the skeleton slot itself, in the packed form of the other seven
    }
    {
        const int nrow = g.pair ? g.W2 : g.W;
        const int srow = g.pair ? (w >> 1) : w;
        int2 v = sv[(size_t)srow * SKK + t];
        if (g.aggx_off) v = sk_fold_aggx(sv[(size_t)g.aggx_off + (size_t)(srow / g.aggx_tpw) * SKK + t], v);
        if (g.pair && (w & 1)) {                            // second tile of its pair: fold the first one's row in (skel_k2_kernel's combine)
            const int2 r0 = (sv + (size_t)nrow * SKK + SKK / 2)[(size_t)(w >> 1) * SKK + t];
            v.y = r0.x ? r0.y : (v.x ? max(v.y, r0.y) : -1);
            v.x += r0.x;
        }
        s_bH[SKK + t] = v.x; s_cH[SKK + t] = v.y; s_tH[SKK + t] = reinterpret_cast<const int *>(sv + (size_t)nrow * SKK)[t];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) { s_raw8[c][t] = 0; s_last8[c][t] = -1; }
    lds_barrier();
    // ballot refinement bit by bit: after bit j-1 the mask of same-j-key lanes
    short rk[EPT][8], pl[EPT][8];                           // [.][j]: rank inside the chunk, previous same-j-key position in the chunk (-1)
    const unsigned long long lt = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int c = r * 4 + wv;
        unsigned long long same = __ballot(key[r] >= 0);
#pragma unroll
        for (int bb = 0; bb < SKB; ++bb) {
            const unsigned long long bal = __ballot((key[r] >> bb) & 1);
            same &= ((key[r] >> bb) & 1) ? bal : ~bal;
            const unsigned long long before = same & lt;
            if (bb < SKB - 1) { rk[r][bb + 1] = (short)__popcll(before); pl[r][bb + 1] = before ? (short)(c * 64 + (63 - __clzll(before))) : (short)-1; }
            else if (key[r] >= 0 && !before) { s_raw8[c][key[r]] = (short)__popcll(same); s_last8[c][key[r]] = (short)(c * 64 + (63 - __clzll(same))); }
        }
    }
    // one barrier per step: sparse-table level l (radix 4: steps 1 .. NL4-1) and, beside it, the fold of level 8-l out of level 9-l
    constexpr int NSTEP = SKB - 1;
#pragma unroll
    for (int l = 1; l <= NSTEP; ++l) {
        lds_barrier();
        if (l < NL4) {
            const int wq = 1 << (2 * (l - 1));              // window of the level below
            const int *lo = TBL(l - 1); int *hi = TBL(l);
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int i = r * BLOCK + t;
                int m = lo[i];
                if (i - wq >= 0) m = max(m, lo[i - wq]);
                if (i - 2 * wq >= 0) m = max(m, lo[i - 2 * wq]);
                if (i - 3 * wq >= 0) m = max(m, lo[i - 3 * wq]);
                hi[i] = m;                                  // windows are clipped at the tile's first position
            }
        }
        const int j = SKB - l;
        if (j >= 1) {
            const int K = 1 << j;
            for (int e = t; e < (NC << j); e += BLOCK) {
                const int c = e >> j, kj = e & (K - 1);
                if (j == SKB - 1) {                          // out of the 8-bit keys' tables (their storage becomes sparse levels >= 2 after this step)
                    s_rawH[c][K + kj] = (short)(s_raw8[c][kj] + s_raw8[c][K + kj]);
                    s_lastH[c][K + kj] = (short)max((int)s_last8[c][kj], (int)s_last8[c][K + kj]);
                } else {
                    s_rawH[c][K + kj] = (short)(s_rawH[c][2 * K + kj] + s_rawH[c][3 * K + kj]);
                    s_lastH[c][K + kj] = (short)max((int)s_lastH[c][2 * K + kj], (int)s_lastH[c][3 * K + kj]);
                }
            }
            if (t < K) {
                const int c0 = s_cH[2 * K + t], c1 = s_cH[3 * K + t];
                s_bH[K + t] = s_bH[2 * K + t] + s_bH[3 * K + t];
                s_tH[K + t] = s_tH[2 * K + t] + s_tH[3 * K + t];
                s_cH[K + t] = (c0 < 0) ? c1 : (c1 < 0) ? c0 : min(c0, c1);   // the later last occurrence has the smaller suffix maximum
            }
        }
    }
    lds_barrier();
    // (i) every level entry (heap index 2..255): exclusive scan over the chunks, in place (count -> base, last -> previous)
    if (t >= 2) {
        int base = 0, last = -1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cn = s_rawH[c][t], lp = s_lastH[c][t];
            s_rawH[c][t] = (short)base; s_lastH[c][t] = (short)last;
            base += cn; if (cn) last = lp;
        }
        if (STAGE) s_loc[t] = (short)base;
    }
    // (ii) per level: bucket bases G (exclusive prefix of the key totals) and the nearest lower non-empty key; one wave per level
    {
        auto level_scan = [&](int j) {
            const int K = 1 << j;
            int carryG = 0, carryL = 0;
            for (int base = 0; base < K; base += 64) {
                const int kj = base + lane;
                const int v = (kj < K) ? s_tH[K + kj] : 0;
                const int ginc = wave_iscan_sum(v), linc = wave_iscan_max(v ? kj + 1 : 0);
                const int lexc = lane_shr1(linc, 0);
                if (kj < K) {
                    const int low = max(carryL, lexc) - 1, c1 = s_cH[K + kj];
                    s_GH[K + kj] = carryG + ginc - v;
                    // what an element without a predecessor in the tile gets, per heap entry instead of per output: the carry (to be
                    // maxed with the range maximum), or the divergence against the nearest lower non-empty key, or 0
                    s_cH[K + kj] = (c1 >= 0) ? (c1 | EFLAG) : (low >= 0) ? k + 1 + (31 - __clz(kj ^ low)) : 0;
                }
                carryG += __builtin_amdgcn_readlane(ginc, 63); carryL = max(carryL, __builtin_amdgcn_readlane(linc, 63));
            }
        };
        if (wv == 0) level_scan(6);
        else if (wv == 1) { level_scan(5); level_scan(1); }
        else if (wv == 2) { level_scan(4); level_scan(2); }
        else { level_scan(3); level_scan(7); }
    }
    lds_barrier();
    if constexpr (STAGE) {
        // (iii) where a bucket starts in the tile's own destination order (Ls, exclusive prefix of the tile's totals per level): s_GH := Ls,
        // s_bH := G + before - Ls, so that local index = Ls + rank in the bucket and destination = local index + s_bH
        auto loc_scan = [&](int j) {
            const int K = 1 << j;
            int carry = 0;
            for (int base = 0; base < K; base += 64) {
                const int kj = base + lane;
                const int v = (kj < K) ? (int)s_loc[K + kj] : 0;
                const int inc = wave_iscan_sum(v);
                if (kj < K) { const int Ls = carry + inc - v, Gb = s_GH[K + kj] + s_bH[K + kj]; s_GH[K + kj] = Ls; s_bH[K + kj] = Gb - Ls; }
                carry += __builtin_amdgcn_readlane(inc, 63);
            }
        };
        if (wv == 0) loc_scan(6);
        else if (wv == 1) { loc_scan(5); loc_scan(1); }
        else if (wv == 2) { loc_scan(4); loc_scan(2); }
        else { loc_scan(3); loc_scan(7); }
        lds_barrier();
        const int nv = min(T, g.M - S);
#pragma unroll
        for (int j = SKB - 1; j >= 1; --j) {
            const int K = 1 << j;
            int *d_out = g.D + (size_t)(8 * b + j) * g.strideD;
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                if (key[r] < 0) continue;
                const int l = r * BLOCK + t, c = r * 4 + wv, kj = key[r] & (K - 1), h = K + kj;
                const int p = (pl[r][j] >= 0) ? pl[r][j] : s_lastH[c][h];
                const int len = l - p, lv = min((31 - __clz(len)) >> 1, NL4 - 1), wq = 1 << (2 * lv);
                const int *tb = TBL(lv);
                const int q3 = p + wq, q1 = max(l - wq, q3), q2 = max(l - 2 * wq, q3);
                const int rm = max(max(tb[l], tb[q3]), max(tb[q1], tb[q2]));
                int dd = rm;
                if (p < 0) { const int e = s_cH[h]; dd = (e & EFLAG) ? max(e & ~EFLAG, rm) : e; }
                const int lp = s_GH[h] + s_rawH[c][h] + rk[r][j];
                s_stage[lp] = dd; s_kb[lp] = (unsigned char)key[r];
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int i = r * BLOCK + t;
                if (i >= nv) continue;
                const int kb = s_kb[i], pos = i + s_bH[K + (kb & (K - 1))];
                int v = s_stage[i];
                if (pos == 0) v = k + j + 1;
#ifdef PBWTAMD_MEASURE
                if (g.dbg_nowrite) continue;
#endif
                if (PACKY == 1) v |= (int)(((unsigned)(kb >> j) & 1u) << 31);
                __builtin_nontemporal_store(v, d_out + pos);
            }
            if (w == g.W - 1 && t == 0) d_out[g.M] = k + j + 1;
            if (j > 1) lds_barrier();
        }
        return;
    }
    // all seven levels, no barrier in between: positions, divergences, scatter
#pragma unroll
    for (int j = SKB - 1; j >= 1; --j) {
        const int K = 1 << j;
        int *a_out = g.A + (size_t)(8 * b + j) * g.strideA, *d_out = g.D + (size_t)(8 * b + j) * g.strideD;
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            if (key[r] < 0) continue;
            const int l = r * BLOCK + t, c = r * 4 + wv, kj = key[r] & (K - 1), h = K + kj;
            const int rank = s_rawH[c][h] + rk[r][j];
            const int p = (pl[r][j] >= 0) ? pl[r][j] : s_lastH[c][h];
            // max d over (p, l]: windows of 4^lv ending at l and at p + 4^lv, and two more in between when the range is longer than 2 / 3 windows
            const int len = l - p, lv = min((31 - __clz(len)) >> 1, NL4 - 1), wq = 1 << (2 * lv);
            const int *tb = TBL(lv);
            const int q3 = p + wq, q1 = max(l - wq, q3), q2 = max(l - 2 * wq, q3);   // windows ending at l, l - wq, l - 2 wq, never starting before p
            const int rm = max(max(tb[l], tb[q3]), max(tb[q1], tb[q2]));
            int dd = rm;
            if (p < 0) { const int e = s_cH[h]; dd = (e & EFLAG) ? max(e & ~EFLAG, rm) : e; }
            const int pos = s_GH[h] + s_bH[h] + rank;
            if (pos == 0) dd = k + j + 1;
#ifdef PBWTAMD_MEASURE
            if (g.dbg_nowrite == 1 && pos >= 0) continue;
            if (g.dbg_nowrite == 2) { __builtin_nontemporal_store(dd, d_out + S + l); continue; }   // same bytes, coalesced, WRONG place: what the scatter itself costs
#endif
            const int yb = (int)(((unsigned)(key[r] >> j) & 1u) << 31);
            // streamed once by the consumers: non-temporal, so the chain's working set stays in L2 (measured +1 %)
            if (PACKY == 1) __builtin_nontemporal_store(dd | yb, d_out + pos);
            else if (PACKY == 2) __builtin_nontemporal_store(dd, d_out + pos);
            else { __builtin_nontemporal_store(av[r] | yb, a_out + pos); __builtin_nontemporal_store(dd, d_out + pos); }
        }
        if (w == g.W - 1 && t == 0) d_out[g.M] = k + j + 1;
    }
}

}  // namespace pbwtk
