// pbwt_k_fillseq.h — the FILL as a tile-local sequential recurrence ("sequential fill").
//
// The seven states between two skeleton states (pbwtCore.c:485-508 applied to sites k+1 .. k+7) are produced beside the chain.  The
// table form (skel_fill_kernel) derives every sub-step independently from the tile's position-ordered data: per-chunk count / last-
// position tables folded down seven times, a sparse table for the range maxima, ~85 VALU instructions and ~10 LDS accesses per output.
// Here ONE WAVE carries its tile through the seven sub-steps in the tile's own sorted order (model: tests/tile_model.py::fillS_tiles):
//   * at level j the tile's elements stand sorted by their j-bit keys; the elements of one key (a "run") are contiguous in the global
//     state k+j as well, so sub-step j -> j+1 is ONE stable partition of the local array by bit j (a prefix count of ones) ...
//   * ... and one segmented max: a "stretch" is a maximal group of neighbours with equal (j+1)-bit key; inside a stretch d' = d (the
//     same-bit predecessor is the neighbour, pbwtCore.c:497-503 with p or q just reset), the head of a stretch takes max(d, max of the
//     stretch before it) when that stretch is not the first of its run; otherwise the element is the FIRST of its (j+1)-bit key in the
//     tile and the folded skeleton tables decide (carry of that key, or the key-difference value) — once per key and tile, not per output;
//   * the tile's elements of one key land at consecutive destinations: every sub-step's outputs leave through LDS in destination order.
// No workgroup barrier anywhere (a wave is its own tile), 4.5 KB of LDS per wave, 2 LDS accesses + the staged store per output.
#pragma once

namespace pbwtk {

constexpr int FS_EFLAG = 0x40000000;                        // ext entry: carry | FS_EFLAG (max with the local maximum) or the final value

// What does not depend on the tile, once per 8-site block: per heap entry h = 2^j + kj (level j = 1 .. 7, kj = the j-bit key)
// {G = first position of the key's bucket in state k+j, base = divergence of an element whose key has no earlier occurrence anywhere:
// k+1+msb(kj ^ nearest lower non-empty key), or 0}.  grid = blocks, 256 threads (thread = 8-bit key).
struct SkFillPrepArgs { const int2 *scan; size_t strideS; int nrow; int kbase; int2 *gb; };
__global__ __launch_bounds__(BLOCK) void skel_fillprep_kernel(SkFillPrepArgs g) {
    __shared__ int s_t[2 * SKK];
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id(), b = blockIdx.x, k = g.kbase + 8 * b;
    s_t[SKK + t] = reinterpret_cast<const int *>(g.scan + (size_t)b * g.strideS + (size_t)g.nrow * SKK)[t];
#pragma unroll
    for (int j = SKB - 1; j >= 1; --j) {
        __syncthreads();
        const int K = 1 << j;
        if (t < K) s_t[K + t] = s_t[2 * K + t] + s_t[3 * K + t];
    }
    __syncthreads();
    int2 *out = g.gb + (size_t)b * SKK;
    auto level_scan = [&](int j) {
        const int K = 1 << j;
        int carryG = 0, carryL = 0;
        for (int base = 0; base < K; base += 64) {
            const int kj = base + lane;
            const int v = (kj < K) ? s_t[K + kj] : 0;
            const int ginc = wave_iscan_sum(v), linc = wave_iscan_max(v ? kj + 1 : 0);
            const int lexc = lane_shr1(linc, 0);
            if (kj < K) {
                const int low = max(carryL, lexc) - 1;
                out[K + kj] = make_int2(carryG + ginc - v, (low >= 0) ? k + 1 + (31 - __clz(kj ^ low)) : 0);
            }
            carryG += __builtin_amdgcn_readlane(ginc, 63); carryL = max(carryL, __builtin_amdgcn_readlane(linc, 63));
        }
    };
    if (wv == 0) level_scan(6);
    else if (wv == 1) { level_scan(5); level_scan(1); }
    else if (wv == 2) { level_scan(4); level_scan(2); }
    else { level_scan(3); level_scan(7); }
}

struct SkFillSeqArgs {
    int *D; size_t strideD;                                 // ring base (slot 0 of the batch part)
    const unsigned char *keys; size_t strideK;              // keys of state 8b at keys + b*strideK
    const int2 *scan; size_t strideS;                       // per block: scan[rows][256] {before, carry}, total[256] (, first halves' rows)
    const int2 *gb;                                         // per block: [256] {G, base} (skel_fillprep_kernel)
    int M, W, kbase, nblk;
    int xcd;                                                // XCD-contiguous workgroups (xcd_tile)
    int pair, W2;                                           // pair rows: scan[W2][256], total, then the first halves' rows tbl0[W2][256] per block
    int aggx_off, aggx_tpw;                                 // as SkFillArgs
#ifdef PBWTAMD_MEASURE
    int dbg_nowrite;
#endif
    // FUSE: the first step of matchMaximalWithin's two scans (pbwtMatch.c:124-129) decided here, and the sites' sorted bit columns emitted
    unsigned *flags; size_t strideF;                        // [site][ceil(M / 32)]: bit p set = position p is NOT decided here (sweep_resid_kernel takes it)
    unsigned long long *ycols; int wpc64;                   // [site][wpc64], zeroed by the caller: the allele column of every state in sorted order (what pack3 encodes)
    unsigned long long *nflag;                              // += positions flagged (the host falls back to the streaming sweep when a panel leaves too many)
    int *Dout;                                              // where the seven filled states (and, PACKY 1, the packed skeleton state) go: D, or the packed-slot ring of the same geometry
    // PACKY 3 (round 4): the 16-BIT hand-off (pbwt_k_common.h: p16_encode).  An escaped d is ALSO stored in the 32-bit ring slot (filled states; a skeleton slot
    // holds its d already).  Half the bytes of the d | y << 31 form on both sides of the hand-off (the fill is bound by its stores).
    unsigned short *P16; size_t stride16; int clip;
};

__device__ __forceinline__ void fs_comb(int &b0, int &c0, int b1, int c1) {      // two keys sharing their low bits: counts add, the later last occurrence has the smaller suffix maximum
    b0 += b1; c0 = (c0 < 0) ? c1 : (c1 < 0) ? c0 : min(c0, c1);
}

// E positions per lane, tile T = 64 E (= the chain's tile: 256 or 512).  PACKY 1: slots get d | y << 31; PACKY 2: d only; PACKY 3: the 16-bit ring (SkFillSeqArgs).
// grid = ceil(W * blocks / 4) workgroups of 4 independent waves; wave -> (block, tile).
//
// FUSE — MEASUREMENT BUILDS ONLY: built, bit-exact, slower (see run_consumers in pbwt_engine.hip for the numbers; kept as the record of the experiment).
// (With PACKY 1, the -stats option set.)  matchMaximalWithin (pbwtMatch.c:115-131) reports position i of a state unless one of its two
// scans meets y[i] again, and almost everywhere the FIRST step decides: with b = y[i], "d[i] <= d[i+1] and y[i-1] == b" or "d[i] >= d[i+1]
// and y[i+1] == b" means not reported.  The tile's elements of one key are neighbours in the state too, so while a level's outputs stand
// in LDS in destination order every position whose neighbours belong to the same run is tested here, from LDS, and only what is left —
// the two ends of every run and the positions whose scans go on — is flagged (one bit per position) for sweep_resid_kernel, which reads
// just those from HBM: on a founder-mosaic panel ~1 % of the positions in ~6 % of the 64-byte lines, instead of every state once more.
// The states' allele columns in sorted order (what pack3 encodes) leave as 64-bit atomic ORs, two per run and 64 positions.
// FUSE 1 = YC (measurement builds, with PACKY 1 and the -stats sweep; bit-exact, a wash: numbers in run_consumers): the fill also EMITS the sorted allele column of every state it writes (and of the skeleton
// state) — the column pack3 encodes and, since round 4, the first thing the sweep reads (sweep_hist_kernel<true, YCIN>): per 64 destination-ordered
// positions one ballot, nothing to do when it is 0 (most sites carry a rare allele), else two 64-bit ORs per stretch of one destination offset.
// FUSE 2 = the measurement-only full fusion described above (decisions + flags as well).
template <int E, int PACKY, int FUSE = 0, int WPE = (E == 8 ? 6 : 4)>
__global__ __launch_bounds__(BLOCK, WPE) void skel_fillseq_kernel(SkFillSeqArgs g) {
    constexpr int T = 64 * E;
    __shared__ __attribute__((aligned(16))) int s_dd[WAVES][T];
    __shared__ __attribute__((aligned(16))) unsigned char s_kk[WAVES][T];
    __shared__ int2 s_tabs[WAVES][SKK];                     // heap-indexed {destination offset, ext}
    const int lane = lane_id(), wv = wave_id_s();    // (scalar: block, tile and every slot pointer below live in SGPRs)
    const int wgl = g.xcd ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    const int lg = wgl * WAVES + wv;
    if (lg >= g.W * g.nblk) return;                         // whole waves only; nothing below synchronises across waves
    const int b = lg / g.W, w = lg - b * g.W;
    int *const s_d = s_dd[wv]; unsigned char *const s_k = s_kk[wv]; int2 *const tab = s_tabs[wv];
    const int S = w * T, k = g.kbase + 8 * b, nvalid = min(T, g.M - S);
    int *const d0 = g.D + (size_t)(8 * b) * g.strideD;
    int *const p0 = g.Dout + (size_t)(8 * b) * g.strideD;       // the skeleton state's slot in the output ring
    const unsigned char *const keys = g.keys + (size_t)b * g.strideK;
    const int2 *const sv = g.scan + (size_t)b * g.strideS;
    const int2 *const gbp = g.gb + (size_t)b * SKK;
    const int l0 = lane * E;

    // ---- the tile: E consecutive positions per lane
    int dc[E]; unsigned kc[E];
    {
        const int4 *dp = reinterpret_cast<const int4 *>(d0 + S + l0);
#pragma unroll
        for (int q = 0; q < E / 4; ++q) { const int4 v = dp[q]; dc[4 * q] = v.x; dc[4 * q + 1] = v.y; dc[4 * q + 2] = v.z; dc[4 * q + 3] = v.w; }
        unsigned kw[E / 4];
        if constexpr (E == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(keys + S + l0); kw[0] = v.x; kw[1] = v.y; }
        else kw[0] = *reinterpret_cast<const unsigned *>(keys + S + l0);
#pragma unroll
        for (int e = 0; e < E; ++e) kc[e] = (kw[e / 4] >> (8 * (e & 3))) & 0xffu;
    }
    // ---- the tile's table rows (issued now, folded below)
    int bq[4], cq[4];
    {
        const int nrow = g.pair ? g.W2 : g.W;
        const int2 *row = sv + (size_t)(g.pair ? (w >> 1) : w) * SKK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int2 v = row[lane + 64 * q];
            if (g.aggx_off) v = sk_fold_aggx(sv[(size_t)g.aggx_off + (size_t)((g.pair ? (w >> 1) : w) / g.aggx_tpw) * SKK + lane + 64 * q], v);
            bq[q] = v.x; cq[q] = v.y;
        }
        if (g.pair && (w & 1)) {                            // second tile of its pair: fold the first one's row in (skel_k2_kernel's combine)
            const int2 *r0p = sv + (size_t)nrow * SKK + SKK / 2 + (size_t)(w >> 1) * SKK;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int2 r0 = r0p[lane + 64 * q]; cq[q] = r0.x ? r0.y : (bq[q] ? max(cq[q], r0.y) : -1); bq[q] += r0.x; }
        }
    }
    int2 gbv[SKB];                                          // [j]: the lane's heap entry of level j (level 7: entries lane and lane + 64 -> gbv[7], gbv[0])
#pragma unroll
    for (int j = 1; j <= 6; ++j) gbv[j] = gbp[(1 << j) + (lane & ((1 << j) - 1))];
    gbv[7] = gbp[128 + lane]; gbv[0] = gbp[192 + lane];

#pragma unroll
    for (int e = 0; e < E; ++e) if (l0 + e >= nvalid) { dc[e] = 0; kc[e] = 0xffu; }    // beyond M: sorts last at every level, never a maximum
    if (PACKY == 3) {                                       // the skeleton slot itself in the 16-bit form (its d stays where it is: escapes read it there)
        unsigned short *const q0 = g.P16 + (size_t)(8 * b) * g.stride16;
        unsigned hv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { bool esc; hv[e] = p16_encode(k, dc[e], kc[e] & 1u, g.clip, esc); }
        if (nvalid == T) {
            if constexpr (E == 8) *reinterpret_cast<uint4 *>(q0 + S + l0) = make_uint4(hv[0] | hv[1] << 16, hv[2] | hv[3] << 16, hv[4] | hv[5] << 16, hv[6] | hv[7] << 16);
            else *reinterpret_cast<uint2 *>(q0 + S + l0) = make_uint2(hv[0] | hv[1] << 16, hv[2] | hv[3] << 16);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) if (l0 + e < nvalid) q0[S + l0 + e] = (unsigned short)hv[e];
        }
        if (w == g.W - 1 && lane == 0) { bool esc; q0[g.M] = (unsigned short)p16_encode(k, d0[g.M], 0u, g.clip, esc); }
    }
    if (PACKY == 1) {                                       // the skeleton slot itself, in the packed form of the other seven
        if (nvalid == T) {
            int4 *dp = reinterpret_cast<int4 *>(p0 + S + l0);
#pragma unroll
            for (int q = 0; q < E / 4; ++q)
                dp[q] = make_int4(dc[4 * q] | (int)((kc[4 * q] & 1u) << 31), dc[4 * q + 1] | (int)((kc[4 * q + 1] & 1u) << 31),
                                  dc[4 * q + 2] | (int)((kc[4 * q + 2] & 1u) << 31), dc[4 * q + 3] | (int)((kc[4 * q + 3] & 1u) << 31));
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) if (l0 + e < nvalid) p0[S + l0 + e] = dc[e] | (int)((kc[e] & 1u) << 31);
        }
        if (g.Dout != g.D && w == g.W - 1 && lane == 0) p0[g.M] = d0[g.M];      // the closing sentinel travels with the state
    }
    // ---- fold the row down the heap, in registers: level 8 keys lane + 64 q; level 7: (q0, q2) -> key lane, (q1, q3) -> key lane + 64;
    // level 6: both; below: xor-shuffles (every lane ends up with the entry of key lane mod 2^j)
    fs_comb(bq[0], cq[0], bq[2], cq[2]); fs_comb(bq[1], cq[1], bq[3], cq[3]);
    tab[128 + lane] = make_int2(gbv[7].x + bq[0], cq[0] >= 0 ? (cq[0] | FS_EFLAG) : gbv[7].y);
    tab[192 + lane] = make_int2(gbv[0].x + bq[1], cq[1] >= 0 ? (cq[1] | FS_EFLAG) : gbv[0].y);
    fs_comb(bq[0], cq[0], bq[1], cq[1]);
    tab[64 + lane] = make_int2(gbv[6].x + bq[0], cq[0] >= 0 ? (cq[0] | FS_EFLAG) : gbv[6].y);
#pragma unroll
    for (int j = 5; j >= 1; --j) {
        const int K = 1 << j;
        const int ob = __shfl_xor(bq[0], K), oc = __shfl_xor(cq[0], K);
        fs_comb(bq[0], cq[0], ob, oc);
        if (lane < K) tab[K + lane] = make_int2(gbv[j].x + bq[0], cq[0] >= 0 ? (cq[0] | FS_EFLAG) : gbv[j].y);
    }
    asm volatile("" ::: "memory");

    int nfl = 0;                                            // FUSE: positions this wave flagged
    // ---- seven sub-steps (FUSE: level 0, the skeleton state itself, first — tested like the others, nothing stored)
    if constexpr (FUSE == 1) {
        // the skeleton state's own allele column (bit 0 of the keys, positions S .. S + T - 1 = whole, tile-aligned words): a lane's E consecutive
        // positions are E bits of one word, OR-ed together over the 64 / E lanes that share it — plain stores, nobody else writes these words
        if (g.ycols) {
            unsigned long long v = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) v |= (unsigned long long)((l0 + e < nvalid) ? (kc[e] & 1u) : 0u) << e;
            v <<= (lane % (64 / E)) * E;
#pragma unroll
            for (int o = 1; o < 64 / E; o <<= 1) v |= __shfl_xor(v, o);
            unsigned long long *const yc0 = g.ycols + (size_t)(8 * b) * g.wpc64 + (S >> 6);
            if ((lane % (64 / E)) == 0 && (S >> 6) + lane / (64 / E) < g.wpc64) yc0[lane / (64 / E)] = v;
        }
    }
#pragma unroll
    for (int j = (FUSE == 2) ? -1 : 0; j < SKB - 1; ++j) {
        const unsigned m1 = (1u << (j + 1)) - 1u, m0 = (j > 0) ? (1u << j) - 1u : 0u;
        // UNIFORM LEVEL (round 4): no element of the tile carries a 1 at site k + j (most sites carry a rare allele: on a founder-mosaic panel
        // two tile-levels in three).  The partition is then the identity (c1 = 0: ni = the element's own index), bit j adds nothing to any key, so
        // every stretch head is the first of its run (H == R) and takes its value from the tables as in the general path, and everything else keeps
        // its d (pbwtCore.c:497-503: the same-allele predecessor is the neighbour).  No prefix count, no segmented max, nothing moves in LDS but the
        // heads' values.
        bool fast = false;
        if constexpr (FUSE == 0) {
            if (j >= 0) {
                unsigned orb = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) orb |= kc[e];
                fast = __ballot((orb >> j) & 1u) == 0ULL;
            }
        }
      if (fast) {
        const unsigned pk = (unsigned)lane_shr1((int)kc[E - 1], 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned x = kc[e] ^ (e ? kc[e - 1] : pk);
            const bool head = ((e == 0) && (lane == 0)) || (x & m0) != 0;
            if (__ballot(head)) {                           // (wave-uniform skip: a tile holds a handful of runs)
                if (head) {                                 // first of its (j+1)-bit key in the tile
                    const int h = (int)(m1 + 1u) + (int)(kc[e] & m1);
                    const int2 tv = tab[h];
                    dc[e] = (tv.y & FS_EFLAG) ? max(tv.y & ~FS_EFLAG, dc[e]) : tv.y;
                    tab[h].x = tv.x - (l0 + e);
                    if (j > 0) s_d[l0 + e] = dc[e];
                }
            }
            if (j == 0) { s_d[l0 + e] = dc[e]; s_k[l0 + e] = (unsigned char)kc[e]; }     // level 0 stages the tile
        }
      } else if (j < 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) { s_d[l0 + e] = dc[e]; s_k[l0 + e] = (unsigned char)kc[e]; }
        if (lane == 0) tab[1] = make_int2(S, 0);
      } else {
        const unsigned pk = (unsigned)lane_shr1((int)kc[E - 1], 0);
        bool H[E], R[E];
        int c1 = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned x = kc[e] ^ (e ? kc[e - 1] : pk);
            const bool first = (e == 0) && (lane == 0);
            H[e] = first || (x & m1) != 0; R[e] = first || (x & m0) != 0;
            c1 += (int)((kc[e] >> j) & 1u);
        }
        const int inc = wave_iscan_sum(c1);
        const int Z = T - __builtin_amdgcn_readlane(inc, 63);
        int p1 = inc - c1;
        // segmented max over the stretches: in the lane, then over the lanes' summaries (fc = 0: no head in the lane; else 1 | 2 * "the stretch starts its run")
        int Sx[E], fx[E];
        int m = 0, fc = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (H[e]) { m = dc[e]; fc = R[e] ? 3 : 1; } else m = max(m, dc[e]);
            Sx[e] = m; fx[e] = fc;
        }
        int sf = fc, sm = m;
#define FS_STEP(CTRL, RM) { const int lf = dpp_mov<CTRL, RM>(0, sf), lm = dpp_mov<CTRL, RM>(0, sm); sm = sf ? sm : max(lm, sm); sf = sf ? sf : lf; }
        FS_STEP(0x111, 0xf) FS_STEP(0x112, 0xf) FS_STEP(0x114, 0xf) FS_STEP(0x118, 0xf) FS_STEP(0x142, 0xa) FS_STEP(0x143, 0xc)
#undef FS_STEP
        const int ef = lane_shr1(sf, 0), em = lane_shr1(sm, 0);      // the open stretch at the end of the previous lane
        int prevS = em, prevf = ef;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int bit = (int)((kc[e] >> j) & 1u);
            const int ni = bit ? Z + p1 : l0 + e - p1;
            p1 += bit;
            int nd = dc[e];
            if (H[e]) {
                const int loc = R[e] ? dc[e] : max(dc[e], prevS);
                if (!R[e] && !(prevf & 2)) nd = loc;
                else {                                      // first of its (j+1)-bit key in the tile
                    const int h = (int)(m1 + 1u) + (int)(kc[e] & m1);
                    const int2 tv = tab[h];
                    nd = (tv.y & FS_EFLAG) ? max(tv.y & ~FS_EFLAG, loc) : tv.y;
                    tab[h].x = tv.x - ni;                   // G + before - (the key's first index in the tile's new order)
                }
            }
            prevS = fx[e] ? Sx[e] : max(Sx[e], em); prevf = fx[e] ? fx[e] : ef;
            s_d[ni] = nd; s_k[ni] = (unsigned char)kc[e];
        }
      }
        asm volatile("" ::: "memory");
        // ---- the level's outputs, destination order: lanes = consecutive local indices = consecutive destinations inside a run
        int *const dout = g.Dout + (size_t)(8 * b + j + 1) * g.strideD;
        // FOUR per lane and store (round 4): a wave64 vector-memory instruction costs the address path its 16 cycles whatever it carries, and the
        // fill was bound by exactly that — 64 stores of 128-256 bytes per wave (without its stores the kernel takes half the time).  A lane takes four
        // consecutive local indices; when they belong to one run (one destination offset — the rule on a panel whose sites carry rare alleles) they
        // leave as ONE store of 8 (16-bit slots; 2-byte aligned: the address mode of this stack takes it) or 16 bytes; a lane at a run boundary
        // stores its four one by one.  Tables and LDS reads shrink alike: one offset lookup, one 4-byte and one 16-byte read per four outputs.
        bool vec_done = false;
        if constexpr (FUSE == 0) {
#ifdef PBWTAMD_MEASURE
          if (!g.dbg_nowrite)
#endif
          if (j >= 0 && nvalid == T) {
            vec_done = true;
            unsigned short *const dout16 = (PACKY == 3) ? g.P16 + (size_t)(8 * b + j + 1) * g.stride16 : nullptr;
            const int site = k + j + 1;
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const int x = 4 * (q * 64 + lane);
                const unsigned k4 = *reinterpret_cast<const unsigned *>(s_k + x);
                const int4 d4 = *reinterpret_cast<const int4 *>(s_d + x);
                int v[4] = {d4.x, d4.y, d4.z, d4.w};
                const unsigned kp[4] = {k4 & m1, (k4 >> 8) & m1, (k4 >> 16) & m1, (k4 >> 24) & m1};
                const unsigned yb[4] = {(k4 >> (j + 1)) & 1u, (k4 >> (j + 9)) & 1u, (k4 >> (j + 17)) & 1u, (k4 >> (j + 25)) & 1u};
                const int off0 = tab[(int)(m1 + 1u) + (int)kp[0]].x;
                // (a wave in which most lanes straddle a run boundary — an iid panel at the deep levels — skips the attempt: both branches would run in full)
                const bool one_run = kp[1] == kp[0] && kp[2] == kp[0] && kp[3] == kp[0];
                const bool try_vec = __popcll(__ballot(one_run)) >= 40;
                if (try_vec && one_run) {
                    const int dest = x + off0;
                    if (dest == 0) v[0] = site + 1;         // sentinel (pbwtCore.c:507)
                    if constexpr (PACKY == 3) {
                        bool e0, e1, e2, e3;
                        const unsigned h0 = p16_encode(site, v[0], yb[0], g.clip, e0), h1 = p16_encode(site, v[1], yb[1], g.clip, e1);
                        const unsigned h2 = p16_encode(site, v[2], yb[2], g.clip, e2), h3 = p16_encode(site, v[3], yb[3], g.clip, e3);
                        struct __attribute__((packed, aligned(2))) U64A2 { unsigned long long u; };
                        reinterpret_cast<U64A2 *>(dout16 + dest)->u = (unsigned long long)(h0 | h1 << 16) | ((unsigned long long)(h2 | h3 << 16) << 32);
                        if (e0 | e1 | e2 | e3) {            // (rare: a match of 32 767 sites or more) the sweep reads d itself from the 32-bit slot
                            if (e0) dout[dest] = v[0];
                            if (e1) dout[dest + 1] = v[1];
                            if (e2) dout[dest + 2] = v[2];
                            if (e3) dout[dest + 3] = v[3];
                        }
                    } else {
                        struct __attribute__((aligned(4))) I128A4 { int x, y, z, w; };     // (16 bytes at a 4-byte aligned address: one global_store_dwordx4)
                        I128A4 t;
                        if constexpr (PACKY == 1) t = I128A4{v[0] | (int)(yb[0] << 31), v[1] | (int)(yb[1] << 31), v[2] | (int)(yb[2] << 31), v[3] | (int)(yb[3] << 31)};
                        else t = I128A4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<I128A4 *>(dout + dest) = t;
                    }
                } else {                                    // a run boundary inside the four
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int dest = x + e + ((e && kp[e] != kp[0]) ? tab[(int)(m1 + 1u) + (int)kp[e]].x : off0);
                        const int vv = (dest == 0) ? site + 1 : v[e];
                        if constexpr (PACKY == 3) {
                            bool esc;
                            const unsigned h = p16_encode(site, vv, yb[e], g.clip, esc);
                            dout16[dest] = (unsigned short)h;
                            if (esc) dout[dest] = vv;
                        } else dout[dest] = (PACKY == 1) ? (vv | (int)(yb[e] << 31)) : vv;
                    }
                }
            }
            if (w == g.W - 1 && lane == 0) { dout[g.M] = k + j + 2; if constexpr (PACKY == 3) dout16[g.M] = 0; }
          }
        }
        if constexpr (FUSE == 0) {
          if (!vec_done && j >= 0) {                        // the panel's ragged last tile: one by one
#ifdef PBWTAMD_MEASURE
            if (!g.dbg_nowrite)
#endif
#pragma nounroll
            for (int q = 0; q < E; ++q) {
                const int x = q * 64 + lane;
                if (x >= nvalid) continue;
                const unsigned kx = s_k[x];
                const int dest = x + tab[(int)(m1 + 1u) + (int)(kx & m1)].x;
                const int vv = (dest == 0) ? k + j + 2 : s_d[x];
                const unsigned yb = (kx >> (j + 1)) & 1u;
                if constexpr (PACKY == 3) {
                    bool esc;
                    const unsigned h = p16_encode(k + j + 1, vv, yb, g.clip, esc);
                    (g.P16 + (size_t)(8 * b + j + 1) * g.stride16)[dest] = (unsigned short)h;
                    if (esc) dout[dest] = vv;
                } else dout[dest] = (PACKY == 1) ? (vv | (int)(yb << 31)) : vv;
            }
            if (w == g.W - 1 && lane == 0) { dout[g.M] = k + j + 2; if constexpr (PACKY == 3) (g.P16 + (size_t)(8 * b + j + 1) * g.stride16)[g.M] = 0; }
          }
        } else {
        int vq[E], pq[E]; unsigned kq[E];
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int x = q * 64 + lane;
            kq[q] = s_k[x];
            pq[q] = x + tab[(int)(m1 + 1u) + (int)(kq[q] & m1)].x;
            vq[q] = s_d[x];
        }
        if (j >= 0) {
            if (pq[0] == 0) vq[0] = k + j + 2;              // sentinel (pbwtCore.c:507): position 0 of the state is the first element of its first run (level 0 holds it already)
#ifdef PBWTAMD_MEASURE
            if (!g.dbg_nowrite)
#endif
            {
                if constexpr (PACKY == 3) {
                    unsigned short *const dout16 = g.P16 + (size_t)(8 * b + j + 1) * g.stride16;
#pragma unroll
                    for (int q = 0; q < E; ++q) {
                        if (q * 64 + lane < nvalid) {
                            bool esc;
                            const unsigned h = p16_encode(k + j + 1, vq[q], (kq[q] >> (j + 1)) & 1u, g.clip, esc);
                            __builtin_nontemporal_store((unsigned short)h, dout16 + pq[q]);
                            if (esc) dout[pq[q]] = vq[q];   // (rare: a match of 32 767 sites or more) the sweep reads d itself from the 32-bit slot
                        }
                    }
                    if (w == g.W - 1 && lane == 0) dout16[g.M] = 0;
                } else
                if (nvalid == T) {                          // (wave-uniform: all tiles but the panel's last)
#pragma unroll
                    for (int q = 0; q < E; ++q) __builtin_nontemporal_store((PACKY == 1) ? (vq[q] | (int)(((kq[q] >> (j + 1)) & 1u) << 31)) : vq[q], dout + pq[q]);
                } else {
#pragma unroll
                    for (int q = 0; q < E; ++q) if (q * 64 + lane < nvalid) __builtin_nontemporal_store((PACKY == 1) ? (vq[q] | (int)(((kq[q] >> (j + 1)) & 1u) << 31)) : vq[q], dout + pq[q]);
                }
            }
            if (w == g.W - 1 && lane == 0) dout[g.M] = k + j + 2;
        }
        if constexpr (FUSE != 0) {
            const int L = j + 1;                            // the state's site is k + L, its alleles bit L of the keys
            unsigned *const fl = (FUSE == 2) ? g.flags + (size_t)(8 * b + L) * g.strideF : nullptr;
            unsigned long long *const yc = g.ycols + (size_t)(8 * b + L) * g.wpc64;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int x = q * 64 + lane;
                const bool valid = x < nvalid;
                const unsigned yI = (kq[q] >> L) & 1u;
                if constexpr (FUSE == 2) {
                // neighbours in the tile's order; beyond the chunk: the next / previous chunk's edge lane; beyond the tile: none
                const int fvr = (q < E - 1) ? __builtin_amdgcn_readlane(vq[q < E - 1 ? q + 1 : q], 0) : 0;
                const int fkr = (q < E - 1) ? __builtin_amdgcn_readlane((int)kq[q < E - 1 ? q + 1 : q], 0) : 0;
                const int fkl = (q > 0) ? __builtin_amdgcn_readlane((int)kq[q > 0 ? q - 1 : q], 63) : 0;
                const int vr = __builtin_amdgcn_update_dpp(fvr, vq[q], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                const unsigned kr = (unsigned)__builtin_amdgcn_update_dpp(fkr, (int)kq[q], 0x130, 0xf, 0xf, false);
                const unsigned kl = (unsigned)lane_shr1((int)kq[q], fkl);
                const bool sameR = ((kr ^ kq[q]) & m1) == 0 && x + 1 < nvalid;      // x + 1 in the tile and in the same run: the state's position + 1
                const bool sameL = ((kl ^ kq[q]) & m1) == 0 && x > 0;
                const unsigned yR = (kr >> L) & 1u, yL = (kl >> L) & 1u;
                const int dI = vq[q], dN = vr;
                const bool skip = sameR && ((dI <= dN && sameL && yL == yI) || (dI >= dN && yR == yI));
                const bool flg = valid && !skip;
                const unsigned long long fm = __ballot(flg);
                if (fm) { nfl += __popcll(fm); if (flg) atomicOr(fl + (pq[q] >> 5), 1u << (pq[q] & 31)); }
                }
                // the allele column: per stretch of lanes with one destination offset, the bits go out as two 64-bit ORs
                const unsigned long long ym = g.ycols ? __ballot(valid && yI) : 0ULL;
                if (ym) {
                    const int offq = pq[q] - x;
                    const int offl = lane_shr1(offq, 0);
                    unsigned long long hm = __ballot(lane == 0 || offl != offq);
                    while (hm) {
                        const int s0 = __ffsll((long long)hm) - 1;
                        hm &= hm - 1;
                        const int s1 = hm ? __ffsll((long long)hm) - 1 : 64;
                        const unsigned long long seg = ((s1 == 64) ? ~0ULL : ((1ULL << s1) - 1ULL)) & ~((1ULL << s0) - 1ULL);
                        const unsigned long long bits = ym & seg;
                        if (!bits) continue;
                        const int base = __builtin_amdgcn_readlane(pq[q], s0) - s0;     // destination of (virtual) lane 0 of this stretch; may be negative
                        const int sh = base & 63, w0 = base >> 6;
                        const unsigned long long lo = bits << sh, hi = sh ? (bits >> (64 - sh)) : 0ULL;
                        if (lane == 0 && lo) atomicOr(yc + w0, lo);
                        if (lane == 1 && hi) atomicOr(yc + w0 + 1, hi);
                    }
                }
            }
        }
        }                                                   // (!vec_done)
        if (j < SKB - 2) {
            const int4 *dp = reinterpret_cast<const int4 *>(s_d + l0);
#pragma unroll
            for (int q = 0; q < E / 4; ++q) { const int4 v = dp[q]; dc[4 * q] = v.x; dc[4 * q + 1] = v.y; dc[4 * q + 2] = v.z; dc[4 * q + 3] = v.w; }
            unsigned kw[E / 4];
            if constexpr (E == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(s_k + l0); kw[0] = v.x; kw[1] = v.y; }
            else kw[0] = *reinterpret_cast<const unsigned *>(s_k + l0);
#pragma unroll
            for (int e = 0; e < E; ++e) kc[e] = (kw[e / 4] >> (8 * (e & 3))) & 0xffu;
        }
        asm volatile("" ::: "memory");
    }
    if constexpr (FUSE == 2) { if (lane == 0 && nfl) atomicAdd(g.nflag, (unsigned long long)nfl); }
}

}  // namespace pbwtk
