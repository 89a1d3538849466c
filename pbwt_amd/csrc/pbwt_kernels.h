// pbwt_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels for the PBWT hot path.
//
// HBM layout:
//   ring slot s:  A[s][Mpad] int32  — prefix array a_k; bits 31/30 carry the alleles of the haplotype at
//                                     that position at the slot's site / the following site ("tags"), a < 2^30
//                 D[s][Mpad+64]     — divergence d_k[0..M] (start-position form, pbwt.h:82, with sentinels)
//   tile summaries                  — per tile of T positions of the NEXT launch's input order, built with
//                                     commutative atomics by the launch that scatters into that order:
//                                     single-site steps: int4 {cnt0, last0+1, last1+1, maxd};
//                                     two-site steps: 3 int4 {c[4]}, {last[4]+1}, {maxd} over the 2-bit keys
//   bit columns  C[k][wpc] uint32   — original order (gather by a) or sorted order (index by position)
//
// Kernels: prepare/prepare2 (tags + summaries of the first site(s) of a pass), step2 (two sites of
// pbwtCursorForwardsA/AD per launch, build side), step1/step (one site per launch: read side, large M),
// and the batch consumers (checksum, maxWithin / longWithin sweeps pbwtMatch.c:85-142, pack3
// encode/decode pbwtCore.c:240-305, query sweep pbwtMatch.c:363-443).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbwtk {

constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int WAVES = BLOCK / 64;
constexpr unsigned TAG = 0x80000000u;
constexpr int AMASK = 0x3fffffff;    // bits 31/30 of a ring entry carry the alleles at the slot's site / the next site

// workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory
// (s_waitcnt vmcnt(0)), which would serialise every barrier behind the outstanding global loads
// and stores this latency-bound kernel deliberately keeps in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave64 cross-lane primitives on DPP (row_shr + row_bcast15/31): a 6-op dependent chain of
// VALU instructions instead of 6 ds_bpermute round trips through the LDS crossbar.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xf, false);
}
#define PBWT_DPP_SCAN(v, OP, ID)                                      \
    v = OP(v, dpp_mov<0x111, 0xf>(ID, v)); /* row_shr:1 */            \
    v = OP(v, dpp_mov<0x112, 0xf>(ID, v)); /* row_shr:2 */            \
    v = OP(v, dpp_mov<0x114, 0xf>(ID, v)); /* row_shr:4 */            \
    v = OP(v, dpp_mov<0x118, 0xf>(ID, v)); /* row_shr:8 */            \
    v = OP(v, dpp_mov<0x142, 0xa>(ID, v)); /* row_bcast:15 */         \
    v = OP(v, dpp_mov<0x143, 0xc>(ID, v)); /* row_bcast:31 */
__device__ __forceinline__ int op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int op_max(int a, int b) { return max(a, b); }
__device__ __forceinline__ int wave_iscan_sum(int v) { PBWT_DPP_SCAN(v, op_add, 0) return v; }
__device__ __forceinline__ int wave_iscan_max(int v) { PBWT_DPP_SCAN(v, op_max, 0) return v; }   // values >= 0
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_iscan_sum(v), 63); }
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_iscan_max(v), 63); }
// value of the previous lane (lane 0 gets `id`)
__device__ __forceinline__ int lane_shr1(int v, int id) {
    return __builtin_amdgcn_update_dpp(id, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ uint64_t sm64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------
// carry tuple of the divergence recurrence (pbwtCore.c:492-503).  For a segment of positions:
//   c0,c1 = number of 0 / 1 alleles; all = max d over the segment;
//   t_b   = max d over the elements after the last allele-b element (all if there is none).
// combine(L,R) is associative; (0,0,0,0,0) is the identity (d >= 0 everywhere).
struct Tup { int c0, c1, t0, t1, all; };

__device__ __forceinline__ Tup tup_combine(const Tup &L, const Tup &R) {
    Tup o;
    o.c0 = L.c0 + R.c0;
    o.c1 = L.c1 + R.c1;
    o.all = max(L.all, R.all);
    o.t0 = R.c0 ? R.t0 : max(L.t0, R.all);
    o.t1 = R.c1 ? R.t1 : max(L.t1, R.all);
    return o;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ Tup tup_dpp(const Tup &v) {      // unwritten lanes get the identity
    Tup r;
    r.c0 = dpp_mov<CTRL, ROWMASK>(0, v.c0); r.c1 = dpp_mov<CTRL, ROWMASK>(0, v.c1);
    r.t0 = dpp_mov<CTRL, ROWMASK>(0, v.t0); r.t1 = dpp_mov<CTRL, ROWMASK>(0, v.t1);
    r.all = dpp_mov<CTRL, ROWMASK>(0, v.all);
    return r;
}
template <bool WITH_D>
__device__ __forceinline__ Tup tup_op(const Tup &L, const Tup &R) {
    if (WITH_D) return tup_combine(L, R);
    return Tup{L.c0 + R.c0, L.c1 + R.c1, 0, 0, 0};
}

// block-wide exclusive scan of Tup over 256 threads (lane order = position order); also returns
// the block total.  smem: WAVES Tups.  One __syncthreads.
template <bool WITH_D>
__device__ __forceinline__ Tup block_scan_tup(Tup v, Tup *smem, Tup &total) {
    const int lane = lane_id(), wv = wave_id();
    Tup inc = v;
    inc = tup_op<WITH_D>(tup_dpp<0x111, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x112, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x114, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x118, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x142, 0xa>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x143, 0xc>(inc), inc);
    if (lane == 63) smem[wv] = inc;
    Tup exc;
    exc.c0 = lane_shr1(inc.c0, 0); exc.c1 = lane_shr1(inc.c1, 0);
    exc.t0 = lane_shr1(inc.t0, 0); exc.t1 = lane_shr1(inc.t1, 0); exc.all = lane_shr1(inc.all, 0);
    lds_barrier();
    Tup pre = Tup{0, 0, 0, 0, 0};
    Tup tot = Tup{0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const Tup sw = smem[w];
        if (w < wv) pre = tup_op<WITH_D>(pre, sw);
        tot = tup_op<WITH_D>(tot, sw);
    }
    total = tot;
    return tup_op<WITH_D>(pre, exc);
}

// ---------------------------------------------------------------------------------------------
// per-batch control block in device memory, written by set_ctl_kernel before each batch so that one
// captured graph serves every batch; read with a single scalar load at kernel start.
struct alignas(32) Ctl {
    int kbase;                 // site index of step 0 of this batch
    int n_total;               // sites in the panel (has_next = k+1 < n_total)
    int pad0, pad1;
    const uint32_t *cols;      // bit columns of this batch: column j = site kbase+j
    const uint32_t *zerocol;   // an all-zero column standing in for sites >= n_total (two-site steps)
};

struct StepArgs {
    const int *a_in;  const int *d_in;     // slot j
    int *a_out;       int *d_out;          // slot j+1
    const Ctl *ctl;
    int4 *summ;                            // [3][wpad] {cnt0,last0+1,last1+1,maxd}; step j reads buffer j%3, accumulates (j+1)%3, clears (j+2)%3
    long long *prof;                       // optional phase timestamps [W][8] (NULL = off)
    int wpc;                               // 32-bit words per column
    int j;                                 // step index inside the batch
    int M, W, wpad;
};

#define PBWT_STAMP(idx) do { if (g.prof && t == 0) g.prof[(size_t)w * 8 + (idx)] = (long long)wall_clock64(); } while (0)

// The step kernel is latency-bound, not bandwidth-bound, for M up to ~1M: one wave
// per SIMD executes its instruction stream exactly once, so the launch time is (instructions on
// the longest path) x (~5 cycles) + the dependent memory round trips.  Hence: one or two positions
// per thread, no validity predication on full tiles (FULL), DPP scans, LDS-only barriers, every
// load whose address is known at entry issued first.
template <int E, bool WITH_D, bool SORTED, bool FULL>
__device__ __forceinline__ void step_body(const StepArgs &g, int *s_a, int *s_d, Tup *s_tup, int (*s_red)[6], int (*s_acc)[4]) {
    constexpr int T = BLOCK * E;
    const int j = g.j;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T;                                   // first position of the tile
    PBWT_STAMP(0);

    const int4 *sm_in = g.summ + (size_t)(j % 3) * g.wpad;
    int4 *sm_out = g.summ + (size_t)((j + 1) % 3) * g.wpad;
    int4 *sm_zero = g.summ + (size_t)((j + 2) % 3) * g.wpad;

    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int av[E], dv[E];
    const int base = S + t * E;                            // blocked: thread t owns E consecutive positions
    if constexpr (E % 4 == 0) {
        const int4 *pa = reinterpret_cast<const int4 *>(g.a_in + base);   // arrays are padded to W*T
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const int4 v = pa[q];
            av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        if (WITH_D) {
            const int4 *pd = reinterpret_cast<const int4 *>(g.d_in + base);
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const int4 v = pd[q];
                dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { av[e] = g.a_in[base + e]; if (WITH_D) dv[e] = g.d_in[base + e]; }
    }
    constexpr int SPT = 4;                                 // summaries per thread (W <= 1024)
    int r_cnt[SPT], r_l0[SPT], r_l1[SPT], r_md[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        r_cnt[q] = 0; r_l0[q] = 0; r_l1[q] = 0; r_md[q] = 0;
        if (jn < W) { const int4 sv = sm_in[jn]; r_cnt[q] = sv.x; r_l0[q] = sv.y; r_l1[q] = sv.z; r_md[q] = sv.w; }
    }
    if (t < 16) s_acc[t >> 2][t & 3] = 0;

    const int k = ctl.kbase + j;
    const bool has_next = (k + 1 < ctl.n_total);           // the panel has a site k+1
    const uint32_t *col_next = ctl.cols + (size_t)(j + 1) * g.wpc;

    // ---- own alleles (tags) and, in gather mode, the next-site allele of each haplotype ----
    unsigned ybits = 0, vbits = FULL ? ((1u << E) - 1u) : 0u, nbits = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned y = ((unsigned)av[e]) >> 31;
        av[e] &= AMASK;
        if (FULL) ybits |= y << e;
        else if (base + e < M) { vbits |= 1u << e; ybits |= y << e; }
    }
    if (!SORTED && has_next) {
        unsigned wd[E];
#pragma unroll
        for (int e = 0; e < E; ++e) wd[e] = (FULL || ((vbits >> e) & 1u)) ? col_next[(unsigned)av[e] >> 5] : 0u;
#pragma unroll
        for (int e = 0; e < E; ++e) nbits |= ((wd[e] >> (av[e] & 31)) & 1u) << e;
    }

    // ---- B: tile summaries of this site -> zero offset, total zeros, last-allele positions ----
    int sumBefore = 0, total = 0, l0 = 0, l1 = 0;
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        total += r_cnt[q];
        if (jn < w) { sumBefore += r_cnt[q]; l0 = max(l0, r_l0[q]); l1 = max(l1, r_l1[q]); }
    }
    sumBefore = wave_sum(sumBefore); total = wave_sum(total);
    if (WITH_D) { l0 = wave_max(l0); l1 = wave_max(l1); }
    if (lane == 0) { s_red[wv][0] = sumBefore; s_red[wv][1] = total; s_red[wv][2] = l0; s_red[wv][3] = l1; }
    lds_barrier();
    sumBefore = 0; total = 0; l0 = 0; l1 = 0;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) {
        sumBefore += s_red[q][0]; total += s_red[q][1];
        l0 = max(l0, s_red[q][2]); l1 = max(l1, s_red[q][3]);
    }
    const int Zw = sumBefore;                              // zeros before this tile
    const int C = total;                                   // zeros in the whole column (u->c)
    PBWT_STAMP(1);

    // carry_b = max d over [l_b, S): the positions after the last allele-b element before the tile
    // = direct reads in the tile holding position l_b - 1, plus whole-tile maxima in between.
    int cw, nvalid;                                        // zeros / valid positions in this tile
    if (WITH_D) {
        int m0 = 0, m1 = 0;
        const int tl0 = l0 ? (l0 - 1) / T : -1, tl1 = l1 ? (l1 - 1) / T : -1;
        int pd0[E], pd1[E];
        const int hi0 = l0 ? min((tl0 + 1) * T, S) : 0, hi1 = l1 ? min((tl1 + 1) * T, S) : 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {                      // issue the dependent loads first
            const int p0 = l0 + t + e * BLOCK, p1 = l1 + t + e * BLOCK;
            pd0[e] = (l0 && p0 < hi0) ? g.d_in[p0] : 0;
            pd1[e] = (l1 && p1 < hi1) ? g.d_in[p1] : 0;
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int jn = t + q * BLOCK;
            if (jn < w) {
                if (l0 && jn > tl0) m0 = max(m0, r_md[q]);
                if (l1 && jn > tl1) m1 = max(m1, r_md[q]);
            }
        }
        // ---- C (overlaps the loads above): thread-local carry tuple ----
        Tup me = Tup{0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                const int d = dv[e];
                me.all = max(me.all, d);
                if (!((ybits >> e) & 1u)) { me.t0 = 0; me.t1 = max(me.t1, d); ++me.c0; }
                else                      { me.t1 = 0; me.t0 = max(me.t0, d); ++me.c1; }
            }
        }
        Tup tot;
        const Tup pre = block_scan_tup<true>(me, s_tup, tot);
        PBWT_STAMP(2);
#pragma unroll
        for (int e = 0; e < E; ++e) { m0 = max(m0, pd0[e]); m1 = max(m1, pd1[e]); }
        m0 = wave_max(m0); m1 = wave_max(m1);
        if (lane == 0) { s_red[wv][4] = m0; s_red[wv][5] = m1; }
        lds_barrier();
        m0 = 0; m1 = 0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { m0 = max(m0, s_red[q][4]); m1 = max(m1, s_red[q][5]); }
        const int carry0 = l0 ? m0 : k + 1;                // nothing before: p starts at k+1 (pbwtCore.c:489)
        const int carry1 = l1 ? m1 : k + 1;
        PBWT_STAMP(3);
        cw = tot.c0; nvalid = tot.c0 + tot.c1;
        int p = pre.c0 ? pre.t0 : max(carry0, pre.all);
        int q1 = pre.c1 ? pre.t1 : max(carry1, pre.all);
        int zi = pre.c0, oi = cw + pre.c1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                int ldst, dn;
                if (!((ybits >> e) & 1u)) { dn = max(p, dv[e]); p = 0; q1 = max(q1, dv[e]); ldst = zi++; }
                else                      { dn = max(q1, dv[e]); q1 = 0; p = max(p, dv[e]); ldst = oi++; }
                s_a[ldst] = av[e] | (int)(((nbits >> e) & 1u) << 31);
                s_d[ldst] = dn;
            }
        }
    } else {
        Tup me = Tup{0, 0, 0, 0, 0};
        me.c0 = __popc(vbits & ~ybits); me.c1 = __popc(vbits & ybits);
        Tup tot;
        const Tup pre = block_scan_tup<false>(me, s_tup, tot);
        PBWT_STAMP(2);
        PBWT_STAMP(3);
        cw = tot.c0; nvalid = tot.c0 + tot.c1;
        int zi = pre.c0, oi = cw + pre.c1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                const int ldst = ((ybits >> e) & 1u) ? oi++ : zi++;
                s_a[ldst] = av[e] | (int)(((nbits >> e) & 1u) << 31);
            }
        }
    }
    lds_barrier();
    PBWT_STAMP(4);

    // ---- D: coalesced write-out in destination order + summaries of site k+1 ----
    const int onesBefore = S - Zw;                         // every earlier tile is full
    const int oneBase = C + onesBefore;                    // destination of this tile's first one
    const int tz = Zw / T, to = oneBase / T;               // first destination tile of each stream
    int mdl[4] = {0, 0, 0, 0};                             // per-lane max d' per destination slot (E > 2)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int l = e * BLOCK + t;
        const bool valid = FULL || (l < nvalid);
        int P = 0, slot = -1, dn = 0;
        unsigned tag = 0;
        if (valid) {
            int a = s_a[l];
            const bool one = l >= cw;
            P = one ? oneBase + (l - cw) : Zw + l;
            slot = one ? 2 + (int)((unsigned)P / T - to) : (int)((unsigned)P / T - tz);
            if (SORTED) {
                if (has_next) tag = (col_next[(unsigned)P >> 5] >> (P & 31)) & 1u;
                a |= (int)(tag << 31);
            } else tag = (unsigned)a >> 31;
            g.a_out[P] = a;
            if (WITH_D) {
                dn = s_d[l];
                if (P == 0) dn = k + 2;                    // sentinel (pbwtCore.c:507)
                g.d_out[P] = dn;
            }
        }
        if (has_next) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned long long mk = __ballot(slot == s);
                if (mk) {                                  // wave-uniform
                    const unsigned long long ones = __ballot(slot == s && tag);
                    const unsigned long long zeros = mk & ~ones;
                    int md = 0;
                    if (WITH_D) {
                        if (E > 2) mdl[s] = max(mdl[s], (slot == s) ? dn : 0);
                        else md = wave_max((slot == s) ? dn : 0);
                    }
                    // lanes of one slot are consecutive positions: P(lane) = P(first) + lane - first
                    const int first = __ffsll((long long)mk) - 1;
                    const int Pf = __builtin_amdgcn_readlane(P, first);
                    if (lane == 0) {
                        atomicAdd(&s_acc[s][0], __popcll(zeros));
                        if (WITH_D) {
                            if (zeros) atomicMax(&s_acc[s][1], Pf + (63 - __clzll(zeros)) - first + 1);
                            if (ones) atomicMax(&s_acc[s][2], Pf + (63 - __clzll(ones)) - first + 1);
                            if (E <= 2) atomicMax(&s_acc[s][3], md);
                        }
                    }
                }
            }
        }
    }
    if (WITH_D && has_next && E > 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = wave_max(mdl[s]);
            if (lane == 0 && m) atomicMax(&s_acc[s][3], m);
        }
    }
    if (WITH_D && w == W - 1 && t == 0) g.d_out[M] = k + 2;
    lds_barrier();
    PBWT_STAMP(5);
    if (has_next && t < 4) {
        const int s = t;
        const int dt = (s < 2 ? tz : to) + (s & 1);
        if (dt < W) {
            int *so = reinterpret_cast<int *>(sm_out + dt);
            const int c0 = s_acc[s][0];
            if (c0) atomicAdd(so, c0);
            if (WITH_D) {
                if (s_acc[s][1]) atomicMax(so + 1, s_acc[s][1]);
                if (s_acc[s][2]) atomicMax(so + 2, s_acc[s][2]);
                if (s_acc[s][3]) atomicMax(so + 3, s_acc[s][3]);
            }
        }
    }
    if (t == 0) sm_zero[w] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(6);
}

// ---------------------------------------------------------------------------------------------
// step2_kernel: TWO sites per launch (gather mode, E = 1).  a_{k+2} is the stable 4-way partition of
// a_k by the key q = b0 | b1<<1 (alleles at sites k, k+1), and every divergence at both levels is a
// static function of (keys, d_k) (tests/tile_model.py::step2_tiles):
//   predecessor with the same key in level-0 order  -> range max of d_k over (pred, e]
//   no such predecessor                             -> k + 1 + msb(q ^ q'), q' = nearest lower non-empty key
//   level-1 value (same b0)                         -> the smaller of the two keys' running maxima
// so the per-launch fixed cost (launch gap + first round trip) is paid once per two sites.
// Tile summaries for the next PAIR of sites, per tile: c[4] (keys), last[4] (+1), maxd — 3 int4,
// all commutative, accumulated by the launch that scatters into that order.
struct Tup4 { int c[4]; int t[4]; int all; };

__device__ __forceinline__ Tup4 tup4_id() { Tup4 r; for (int q = 0; q < 4; ++q) { r.c[q] = 0; r.t[q] = 0; } r.all = 0; return r; }
__device__ __forceinline__ Tup4 tup4_combine(const Tup4 &L, const Tup4 &R) {
    Tup4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) { o.c[q] = L.c[q] + R.c[q]; o.t[q] = R.c[q] ? R.t[q] : max(L.t[q], R.all); }
    o.all = max(L.all, R.all);
    return o;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ Tup4 tup4_dpp(const Tup4 &v) {
    Tup4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.c[q] = dpp_mov<CTRL, ROWMASK>(0, v.c[q]); r.t[q] = dpp_mov<CTRL, ROWMASK>(0, v.t[q]); }
    r.all = dpp_mov<CTRL, ROWMASK>(0, v.all);
    return r;
}
// wave totals -> every wave redundantly scans them in its first NW lanes (NW <= 16: one DPP row),
// so a block of up to 1024 threads needs a single barrier and no per-thread loop over the waves
// part 1 (before the barrier): per-wave inclusive scan, wave total to LDS; part 2 (after the barrier)
// finishes.  Split so that a caller can post other per-wave results under the same barrier.
__device__ __forceinline__ Tup4 wave_scan_tup4(Tup4 v, Tup4 *smem, Tup4 &exc) {
    const int lane = lane_id(), wv = wave_id();
    Tup4 inc = v;
    inc = tup4_combine(tup4_dpp<0x111, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x112, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x114, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x118, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x142, 0xa>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x143, 0xc>(inc), inc);
    if (lane == 63) smem[wv] = inc;
#pragma unroll
    for (int q = 0; q < 4; ++q) { exc.c[q] = lane_shr1(inc.c[q], 0); exc.t[q] = lane_shr1(inc.t[q], 0); }
    exc.all = lane_shr1(inc.all, 0);
    return inc;
}
template <int NW>
__device__ __forceinline__ Tup4 block_scan_finish_tup4(const Tup4 &exc, const Tup4 *smem, Tup4 &total) {
    const int lane = lane_id(), wv = wave_id();
    Tup4 wt = tup4_id();
    if (lane < NW) wt = smem[lane];
    wt = tup4_combine(tup4_dpp<0x111, 0xf>(wt), wt);
    wt = tup4_combine(tup4_dpp<0x112, 0xf>(wt), wt);
    if (NW > 4) { wt = tup4_combine(tup4_dpp<0x114, 0xf>(wt), wt); wt = tup4_combine(tup4_dpp<0x118, 0xf>(wt), wt); }
    Tup4 pre = tup4_id(), tot;
    const int src = (wv > 0) ? wv - 1 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = __builtin_amdgcn_readlane(wt.c[q], src), pt = __builtin_amdgcn_readlane(wt.t[q], src);
        if (wv > 0) { pre.c[q] = pc; pre.t[q] = pt; }
        tot.c[q] = __builtin_amdgcn_readlane(wt.c[q], NW - 1); tot.t[q] = __builtin_amdgcn_readlane(wt.t[q], NW - 1);
    }
    { const int pa = __builtin_amdgcn_readlane(wt.all, src); if (wv > 0) pre.all = pa; tot.all = __builtin_amdgcn_readlane(wt.all, NW - 1); }
    total = tot;
    return tup4_combine(pre, exc);
}

// combine one value per wave across the block (first NW lanes of every wave reduce the NW wave values)
template <int NW, bool IS_MAX>
__device__ __forceinline__ int waves_combine(const int *col /* stride 16 ints per wave */, int lane) {
    int v = (lane < NW) ? col[lane * 16] : 0;
    if (IS_MAX) {
        v = max(v, dpp_mov<0x111, 0xf>(0, v)); v = max(v, dpp_mov<0x112, 0xf>(0, v));
        if (NW > 4) { v = max(v, dpp_mov<0x114, 0xf>(0, v)); v = max(v, dpp_mov<0x118, 0xf>(0, v)); }
    } else {
        v += dpp_mov<0x111, 0xf>(0, v); v += dpp_mov<0x112, 0xf>(0, v);
        if (NW > 4) { v += dpp_mov<0x114, 0xf>(0, v); v += dpp_mov<0x118, 0xf>(0, v); }
    }
    return __builtin_amdgcn_readlane(v, NW - 1);
}

struct Step2Args {
    const int *a_in; const int *d_in;      // slot 2*jl   (state before site k = kbase + 2*jl)
    int *a_mid; int *d_mid;                // slot 2*jl+1 (before site k+1)
    int *a_out; int *d_out;                // slot 2*jl+2 (before site k+2)
    const Ctl *ctl;
    int4 *summ;                            // [3][wpad][3] int4: {c[4]}, {last[4]}, {maxd,0,0,0}
    long long *prof;
    int wpc, jl, M, W, wpad;
};

// NT threads per workgroup, E consecutive positions per thread: tile of T = NT*E positions
// (NT=256,E=1 for M <= 262144; NT=256,E=4 up to M = 1048576, all tiles resident at once).
template <bool WITH_D, bool FULL, int SPT, int NT, int E>
__device__ __forceinline__ void step2_body(const Step2Args &g, Tup4 *s_tup, int (*s_red)[16], int *s_acc) {
    constexpr int T = NT * E, NW = NT / 64;
    const int jl = g.jl;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T, i0 = S + t * E;
    PBWT_STAMP(0);
    const int4 *sm_in = g.summ + (size_t)(jl % 3) * g.wpad * 3;
    int4 *sm_out = g.summ + (size_t)((jl + 1) % 3) * g.wpad * 3;
    int4 *sm_zero = g.summ + (size_t)((jl + 2) % 3) * g.wpad * 3;
    if (t < 72) s_acc[t] = 0;

    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int a[E], d[E];
    if constexpr (E == 4) {
        const int4 va = *reinterpret_cast<const int4 *>(g.a_in + i0);
        a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
        if (WITH_D) { const int4 vd = *reinterpret_cast<const int4 *>(g.d_in + i0); d[0] = vd.x; d[1] = vd.y; d[2] = vd.z; d[3] = vd.w; }
        else { d[0] = d[1] = d[2] = d[3] = 0; }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { a[e] = g.a_in[i0 + e]; d[e] = WITH_D ? g.d_in[i0 + e] : 0; }
    }
    int4 sc[SPT], sl[SPT]; int smx[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * NT;
        sc[q] = make_int4(0, 0, 0, 0); sl[q] = make_int4(0, 0, 0, 0); smx[q] = 0;
        if (jn < W) { sc[q] = sm_in[(size_t)jn * 3]; if (WITH_D) { sl[q] = sm_in[(size_t)jn * 3 + 1]; smx[q] = sm_in[(size_t)jn * 3 + 2].x; } }
    }
    const int k = ctl.kbase + 2 * jl;
    // alleles of the owned haplotypes at sites k+2, k+3: the tags of slot 2*jl+2 = the next launch's keys
    const uint32_t *c2 = (k + 2 < ctl.n_total) ? ctl.cols + (size_t)(2 * jl + 2) * g.wpc : ctl.zerocol;
    const uint32_t *c3 = (k + 3 < ctl.n_total) ? ctl.cols + (size_t)(2 * jl + 3) * g.wpc : ctl.zerocol;
    int key[E], nkey[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        valid[e] = FULL || (i0 + e < M);
        key[e] = (int)(((unsigned)a[e] >> 31) | (((unsigned)a[e] >> 29) & 2u));     // b0 | b1<<1
        a[e] &= AMASK;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        nkey[e] = 0;
        if (valid[e]) nkey[e] = (int)(((c2[(unsigned)a[e] >> 5] >> (a[e] & 31)) & 1u) | (((c3[(unsigned)a[e] >> 5] >> (a[e] & 31)) & 1u) << 1));
    }

    // ---- fold the tile summaries ----
    int bef[4] = {0, 0, 0, 0}, tot4[4] = {0, 0, 0, 0}, lst[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * NT;
        const int cc[4] = {sc[q].x, sc[q].y, sc[q].z, sc[q].w};
        const int ll[4] = {sl[q].x, sl[q].y, sl[q].z, sl[q].w};
#pragma unroll
        for (int x = 0; x < 4; ++x) { tot4[x] += cc[x]; if (jn < w) { bef[x] += cc[x]; lst[x] = max(lst[x], ll[x]); } }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) { bef[x] = wave_iscan_sum(bef[x]); tot4[x] = wave_iscan_sum(tot4[x]); if (WITH_D) lst[x] = wave_iscan_max(lst[x]); }
    if (lane == 63) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { s_red[wv][x] = bef[x]; s_red[wv][4 + x] = tot4[x]; s_red[wv][8 + x] = lst[x]; }
    }
    lds_barrier();
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        bef[x] = waves_combine<NW, false>(&s_red[0][x], lane);
        tot4[x] = waves_combine<NW, false>(&s_red[0][4 + x], lane);
        lst[x] = WITH_D ? waves_combine<NW, true>(&s_red[0][8 + x], lane) : 0;
    }
    PBWT_STAMP(1);
    // carries: max d_k over [last[x], S) = whole-tile maxima + one partial-tile read (<= T positions) per key
    int mx[4] = {0, 0, 0, 0};
    int pd[4][E];
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int tl = lst[x] ? (lst[x] - 1) / T : -1;
            const int hi = lst[x] ? min((tl + 1) * T, S) : 0;
#pragma unroll
            for (int e = 0; e < E; ++e) { const int p = lst[x] + t + e * NT; pd[x][e] = (p < hi) ? g.d_in[p] : 0; }
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int jn = t + q * NT; if (jn < w && jn > tl) mx[x] = max(mx[x], smx[q]); }
        }
    }

    // ---- the thread's own tuple (E positions in order), block scan ----
    Tup4 me = tup4_id();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (valid[e]) {
#pragma unroll
            for (int x = 0; x < 4; ++x) { if (x == key[e]) { ++me.c[x]; me.t[x] = 0; } else me.t[x] = max(me.t[x], d[e]); }
            me.all = max(me.all, d[e]);
        }
    }
    Tup4 tot, exc;
    wave_scan_tup4(me, s_tup, exc);
    PBWT_STAMP(2);
    // the carries' per-wave maxima ride on the scan's barrier (the partial-tile loads had the scan to land)
    int cr[4] = {0, 0, 0, 0};
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
#pragma unroll
            for (int e = 0; e < E; ++e) mx[x] = max(mx[x], pd[x][e]);
            mx[x] = wave_iscan_max(mx[x]);
        }
        if (lane == 63) {
#pragma unroll
            for (int x = 0; x < 4; ++x) s_red[wv][12 + x] = mx[x];
        }
    }
    lds_barrier();
    Tup4 run = block_scan_finish_tup4<NW>(exc, s_tup, tot);   // exclusive prefix of this thread's first position
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) cr[x] = waves_combine<NW, true>(&s_red[0][12 + x], lane);
    }
    PBWT_STAMP(3);
    // ---- per position: divergences and destinations at both levels, scatter, next-pair summaries ----
    const int Zw1 = bef[0] + bef[2], C1 = tot4[0] + tot4[2];
    int G2[4]; G2[0] = 0; G2[1] = tot4[0]; G2[2] = tot4[0] + tot4[1]; G2[3] = tot4[0] + tot4[1] + tot4[2];
    const bool has_next = (k + 2 < ctl.n_total);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (valid[e]) {
            const int ky = key[e], de = d[e];
            int dd1 = 0, dd2 = 0;
            if (WITH_D) {
                int eff[4]; bool ex[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { ex[x] = run.c[x] || lst[x]; eff[x] = run.c[x] ? run.t[x] : max(cr[x], run.all); }
                int e2 = 0; bool x2 = false;
#pragma unroll
                for (int x = 0; x < 4; ++x) if (x == ky) { e2 = eff[x]; x2 = ex[x]; }
                if (x2) dd2 = max(e2, de);                 // same key: range max of d_k since that predecessor
                else {                                     // first of its key: differs from the last element of the nearest lower non-empty key
                    int lower = -1;
#pragma unroll
                    for (int x = 0; x < 4; ++x) if (x < ky && tot4[x] > 0) lower = x;
                    dd2 = (lower >= 0) ? k + 1 + (31 - __clz(ky ^ lower)) : 0;
                }
                // level 1: same allele at site k = the later of the two keys sharing b0 = the smaller maximum
                const int bb = ky & 1;
                const int ea = bb ? eff[1] : eff[0], eb = bb ? eff[3] : eff[2];
                const bool xa = bb ? ex[1] : ex[0], xb = bb ? ex[3] : ex[2];
                dd1 = (xa || xb) ? max(min(xa ? ea : 0x7fffffff, xb ? eb : 0x7fffffff), de) : k + 1;
            }
            const int b0 = ky & 1, b1 = ky >> 1;
            const int zr = run.c[0] + run.c[2], orr = run.c[1] + run.c[3];
            int prk = 0, base2 = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) if (x == ky) { base2 = G2[x] + bef[x]; prk = run.c[x]; }
            const int av1 = a[e] | (int)((unsigned)b1 << 31);
            const int av2 = a[e] | (int)(((unsigned)(nkey[e] & 1) << 31) | ((unsigned)(nkey[e] >> 1) << 30));
            {
                const int pos1 = b0 ? C1 + (S - Zw1) + orr : Zw1 + zr;
                const int pos2 = base2 + prk;
                g.a_mid[pos1] = av1;
                g.a_out[pos2] = av2;
                if (WITH_D) {
                    g.d_mid[pos1] = pos1 ? dd1 : k + 2;    // sentinels (pbwtCore.c:507)
                    if (pos2 == 0) dd2 = k + 3;
                    g.d_out[pos2] = dd2;
                }
                if (has_next) {                            // <= 2 destination tiles per key stream
                    const int slot = ky * 2 + (pos2 / T - base2 / T);
                    atomicAdd(&s_acc[slot * 9 + nkey[e]], 1);
                    if (WITH_D) { atomicMax(&s_acc[slot * 9 + 4 + nkey[e]], pos2 + 1); atomicMax(&s_acc[slot * 9 + 8], dd2); }
                }
            }
            if (E > 1) {                                   // advance the running prefix past this position
#pragma unroll
                for (int x = 0; x < 4; ++x) { if (x == ky) { ++run.c[x]; run.t[x] = 0; } else run.t[x] = max(run.t[x], de); }
                run.all = max(run.all, de);
            }
        }
    }
    if (WITH_D && w == W - 1 && t == 0) { g.d_mid[M] = k + 2; g.d_out[M] = k + 3; }
    PBWT_STAMP(4);
    if (has_next) {
        lds_barrier();
        if (t < 72) {
            const int slot = t / 9, f = t - slot * 9, kq = slot >> 1;
            const int fq = (G2[kq] + bef[kq]) / T;         // first destination tile of stream kq
            const int dt = fq + (slot & 1);
            const int v = s_acc[t];
            if (v && dt < W) {
                int *so = reinterpret_cast<int *>(sm_out + (size_t)dt * 3) + f;
                if (f < 4) atomicAdd(so, v); else atomicMax(so, v);
            }
        }
    }
    if (t < 3) sm_zero[(size_t)w * 3 + t] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(5);
    PBWT_STAMP(6);
}

template <bool WITH_D, int SPT, int NT, int E>
__global__ __launch_bounds__(NT) void step2_kernel(Step2Args g) {
    __shared__ Tup4 s_tup[NT / 64];
    __shared__ int s_red[NT / 64][16];
    __shared__ int s_acc[72];
    if ((int)(blockIdx.x + 1) * NT * E <= g.M) step2_body<WITH_D, true, SPT, NT, E>(g, s_tup, s_red, s_acc);
    else step2_body<WITH_D, false, SPT, NT, E>(g, s_tup, s_red, s_acc);
}

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
};

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
template <int EPT, bool HALF>
__device__ __forceinline__ void skel_hist_body(const SkArgs &g) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);                          // the dependent chain shares SIMDs with the throughput kernels of the consumer stream: issue first
#endif
    constexpr int T = BLOCK * EPT;
    __shared__ int h_cnt[SKK], h_last[SKK];
    __shared__ int s_suf[T];
    __shared__ int s_w[WAVES];
    __shared__ int h_cnt0[HALF ? SKK : 1], h_last0[HALF ? SKK : 1], s_suf0[HALF ? T / 2 : 1];
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id(), w = g.w0 + ((g.xcd & 4) ? xcd_tile(blockIdx.x, g.W) : blockIdx.x);
    const int rb = BLOCK - 1 - t;
    const int l0 = rb * EPT, i0 = w * T + l0;
    unsigned packed;
    int dv[EPT];
    if constexpr (EPT == 4) {
        packed = *reinterpret_cast<const unsigned *>(g.keys + i0);
        const int4 vd = *reinterpret_cast<const int4 *>(g.d + i0);
        dv[0] = vd.x; dv[1] = vd.y; dv[2] = vd.z; dv[3] = vd.w;
    } else if constexpr (EPT == 2) {
        packed = *reinterpret_cast<const unsigned short *>(g.keys + i0);
        const int2 vd = *reinterpret_cast<const int2 *>(g.d + i0);
        dv[0] = vd.x; dv[1] = vd.y;
    } else {
        packed = g.keys[i0]; dv[0] = g.d[i0];
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
    const int c = h_cnt[t], tl = c ? s_suf[h_last[t]] : tilemax;
    g.tbl[(size_t)w * SKK + t] = make_int2(c, tl);         // row-major: one coalesced 2 KB row per tile
    if (HALF) {
        const int c0 = h_cnt0[t], tl0 = c0 ? s_suf0[h_last0[t]] : max(s_w[2], s_w[3]);
        g.tbl0[(size_t)w * SKK + t] = make_int2(c0, tl0);
    }
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
template <int KPW, int TPL>
__device__ __forceinline__ void skel_k2_body(const Sk2Args &g) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int NT = KPW * 64;                            // one wave per key
    constexpr int WP = 64 * (TPL + 1);                      // a lane's TPL tiles + one pad entry: lane stride TPL+1 is odd, no LDS bank conflicts
    __shared__ int2 s_v[KPW][WP];
    const int t = threadIdx.x, lane = lane_id(), kk = t >> 6, key0 = blockIdx.x * KPW;
    constexpr int NIT = 64 * TPL * KPW / NT;                // = TPL: all loads in flight at once (one round trip, not NIT)
    int2 ld[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * NT, r = idx / KPW, kq = idx % KPW;
        ld[i] = (r < g.W) ? g.tbl[(size_t)r * SKK + key0 + kq] : make_int2(0, 0);
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
        if (lane == 63) g.total[key0 + kk] = ic;
    }
    __syncthreads();
    for (int idx = t; idx < g.W * KPW; idx += NT) {
        const int r = idx / KPW, kq = idx % KPW;
        g.scan[(size_t)r * SKK + key0 + kq] = s_v[kq][r + r / TPL];
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
struct Sk2WArgs { const int2 *tbl; int2 *scan; int *total; int W; unsigned long long *agg; unsigned *counter; unsigned target; int *err; };
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
__device__ __forceinline__ void shard_wait_flags(const unsigned *mine, const unsigned *perr, int n, unsigned epoch, int *err, int code) {
    const int t = threadIdx.x;
    if (t < n) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        long spins = 0;
        while ((int)(__hip_atomic_load(mine + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if ((spins & 1023) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                         __hip_atomic_load(perr + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) { atomicCAS(err, 0, 9); break; }
            if (spins > (1L << 24)) { atomicCAS(err, 0, code); break; }
        }
    }
}
// which: 0 = f1, 1 = f2, 2 = f3.  mode bit 0: signal every rank (this one included), bit 1: wait for every rank
__global__ __launch_bounds__(64) void shard_xbar_kernel(ShardPeers P, int which, int mode, unsigned epoch, int *err) {
    const int t = threadIdx.x;
    if (t < P.n && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)       // this rank has failed: tell every peer, so that none waits out its own timeout
        __hip_atomic_store(&P.x[t]->perr[P.me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((mode & 1) && t < P.n) {
        unsigned *f = which == 0 ? P.x[t]->f1 : which == 1 ? P.x[t]->f2 : P.x[t]->f3;
        __hip_atomic_store(f + P.me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
template <int EPT, int TR, bool R4, bool SHARD>
__device__ __forceinline__ void skel_rank_body(const SkArgs &g, const SkShardOut *so) {
#ifndef PBWT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);                          // the dependent chain shares SIMDs with the throughput kernels of the consumer stream: issue first
#endif
    constexpr int T = BLOCK * EPT, NC = EPT * WAVES;        // positions per tile, 64-position chunks per tile
    constexpr int NL = R4 ? ((EPT == 1) ? 4 : 5) : ((EPT == 4) ? 10 : (EPT == 2) ? 9 : 8);   // sparse table levels: windows 1 .. T/2 (radix 2) or 1 .. 4^(NL-1) (radix 4)
    __shared__ short s_cnt[NC][SKK];                        // per chunk: count -> base (exclusive over chunks)
    __shared__ short s_lastp[NC][SKK];                      // per chunk: last local position of the key -> previous one before the chunk
    __shared__ int s_tbl[NL][T];                            // s_tbl[l][i] = max d over (i-2^l, i]
    __shared__ int s_before[SKK], s_carry[SKK], s_G[SKK], s_lower[SKK];
    __shared__ int s_gw[WAVES], s_lw[WAVES];
    __shared__ int *s_pa[SHARD ? SHARD_MAX : 1], *s_pd[SHARD ? SHARD_MAX : 1]; __shared__ unsigned char *s_pk[SHARD ? SHARD_MAX : 1];
    __shared__ int s_pb[SHARD ? SHARD_MAX : 1];
    __shared__ int s_failed;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id(), w = g.w0 + ((g.xcd & 2) ? xcd_tile(blockIdx.x, g.W) : blockIdx.x);
    const int S = w * T;
    if constexpr (SHARD) { if (t == 0) s_failed = __hip_atomic_load(so->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // read by all after the first barrier
    if constexpr (SHARD) { if (t < SHARD_MAX) { s_pa[t] = so->a[t]; s_pd[t] = so->d[t]; s_pk[t] = so->k[t]; s_pb[t] = so->pb[t + 1]; } }   // visible after the barriers below
    int av[EPT], dv[EPT], key[EPT];
    unsigned nk[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {                         // striped: chunk r*4+wv = 64 consecutive positions
        const int i = S + r * BLOCK + t;
        av[r] = g.a[i]; dv[r] = g.d[i]; key[r] = (int)g.keys[i];
    }
    int2 row[TR > 0 ? TR : 1];
    int bq = 0, cq = -1, tq = 0;
    if constexpr (TR > 0) {
#pragma unroll
        for (int r = 0; r < TR; ++r) row[r] = (r < g.W) ? g.tbl[(size_t)r * SKK + t] : make_int2(0, 0);
    } else {
        int2 sv = g.scan[(size_t)(g.pair ? (w >> 1) : w) * SKK + t];
        if (g.pair && (w & 1)) {                            // second tile of its pair: fold the first one's row in
            const int2 r0 = g.tbl0[(size_t)(w >> 1) * SKK + t];
            sv.y = r0.x ? r0.y : (sv.x ? max(sv.y, r0.y) : -1);
            sv.x += r0.x;
        }
        bq = sv.x; cq = sv.y; tq = g.total[t];
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
                s_tbl[l][i] = m;
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
    if (lane == 63) { s_gw[wv] = ginc; s_lw[wv] = linc; }
    const int lexc = lane_shr1(linc, 0);
    lds_barrier();
    int Gq = ginc - tq, lq = lexc;
    for (int x = 0; x < wv; ++x) { Gq += s_gw[x]; lq = max(lq, s_lw[x]); }
    lq -= 1;
    s_before[t] = bq; s_carry[t] = cq; s_G[t] = Gq; s_lower[t] = lq;
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
        else if (s_carry[ky] >= 0) dd = max(s_carry[ky], rm);
        else if (s_lower[ky] >= 0) dd = g.k + 1 + (31 - __clz(ky ^ s_lower[ky]));
        else dd = 0;
        const int pos = s_G[ky] + s_before[ky] + rank;
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
            g.a_out[pos] = av[r] | (int)((nk[r] & 1u) << 31);
            g.d_out[pos] = dd;
            g.keys_out[pos] = (unsigned char)nk[r];
        }
    }
    if (w == g.Wtot - 1 && t == 0) {
        if constexpr (SHARD) so->d[so->n - 1][g.M] = g.k + SKB + 1;      // d[M] lives with the last rank
        else g.d_out[g.M] = g.k + SKB + 1;
    }
}
template <int EPT, int TR, bool R4 = false>
__global__ __launch_bounds__(BLOCK) void skel_rank_kernel(SkArgs g) { skel_rank_body<EPT, TR, R4, false>(g, nullptr); }
// position-sharded form: tiles w0 .. w0+W-1, scatter through the owners' table
template <int EPT, bool R4>
__global__ __launch_bounds__(BLOCK) void skel_rank_shard_kernel(SkArgs g, SkShardOut so) { skel_rank_body<EPT, 0, R4, true>(g, &so); }

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
template <int EPT, int TR>
__global__ __launch_bounds__(BLOCK) void skel_rank_many_kernel(const SkArgs *args) { const SkArgs g = args[blockIdx.y]; skel_rank_body<EPT, TR, false, false>(g, nullptr); }

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
        int2 v = sv[(size_t)(g.pair ? (w >> 1) : w) * SKK + t];
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
#include "pbwt_fillseq.h"
namespace pbwtk {

// first pair of a pass (or after an odd-length batch): both allele tags of slot 0 from columns k, k+1
// and the pair summaries from scratch; clears the accumulation buffer of the first launch
struct Prep2Args { int *a; const int *d; const uint32_t *col0; const uint32_t *col1; int4 *summ; int M, W, wpad, with_d, T; };
__global__ __launch_bounds__(BLOCK) void prepare2_kernel(Prep2Args g) {
    __shared__ int s_acc[9];
    const int t = threadIdx.x, w = blockIdx.x;
    if (t < 9) s_acc[t] = 0;
    __syncthreads();
    for (int i = w * g.T + t; i < min((w + 1) * g.T, g.M); i += BLOCK) {
        const int a = g.a[i] & AMASK;
        const unsigned b0 = (g.col0[(unsigned)a >> 5] >> (a & 31)) & 1u, b1 = (g.col1[(unsigned)a >> 5] >> (a & 31)) & 1u;
        g.a[i] = a | (int)((b0 << 31) | (b1 << 30));
        const int key = (int)(b0 | (b1 << 1));
        atomicAdd(&s_acc[key], 1);
        atomicMax(&s_acc[4 + key], i + 1);
        if (g.with_d) atomicMax(&s_acc[8], g.d[i]);
    }
    __syncthreads();
    if (t == 0) {
        g.summ[(size_t)w * 3] = make_int4(s_acc[0], s_acc[1], s_acc[2], s_acc[3]);
        g.summ[(size_t)w * 3 + 1] = make_int4(s_acc[4], s_acc[5], s_acc[6], s_acc[7]);
        g.summ[(size_t)w * 3 + 2] = make_int4(s_acc[8], 0, 0, 0);
        int4 *nxt = g.summ + (size_t)g.wpad * 3;
        nxt[(size_t)w * 3] = make_int4(0, 0, 0, 0); nxt[(size_t)w * 3 + 1] = make_int4(0, 0, 0, 0); nxt[(size_t)w * 3 + 2] = make_int4(0, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// step1_kernel: the E = 1 specialisation (T = 256, M <= 262144), written for the shortest
// instruction stream: SPT = ceil(W/256) summaries per thread, results scattered straight from
// registers (no LDS staging: the zeros of a wave go to one contiguous destination range, the ones to
// another), every wave posts its own next-site summaries with global atomics.  Three LDS barriers.
template <bool WITH_D, bool SORTED, bool FULL, int SPT>
__device__ __forceinline__ void step1_body(const StepArgs &g, int *s_a, int *s_d, Tup *s_tup, int (*s_red)[6], int (*s_acc)[4]) {
    constexpr int T = BLOCK;
    const int j = g.j;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T, i = S + t;
    PBWT_STAMP(0);
    const int4 *sm_in = g.summ + (size_t)(j % 3) * g.wpad;
    int4 *sm_out = g.summ + (size_t)((j + 1) % 3) * g.wpad;
    int4 *sm_zero = g.summ + (size_t)((j + 2) % 3) * g.wpad;

    if (t < 16) s_acc[t >> 2][t & 3] = 0;
    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int a = g.a_in[i];                                     // padded to W*T
    int d = WITH_D ? g.d_in[i] : 0;
    int4 sv[SPT];                                          // the W tile summaries, SPT per thread
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        sv[q] = (jn < W) ? sm_in[jn] : make_int4(0, 0, 0, 0);
    }
    const int k = ctl.kbase + j;
    const bool has_next = (k + 1 < ctl.n_total);
    const uint32_t *col_next = ctl.cols + (size_t)(j + 1) * g.wpc;
    const bool valid = FULL || (i < M);
    const unsigned y = ((unsigned)a) >> 31;
    a &= AMASK;
    unsigned nbit = 0;
    if (!SORTED && has_next && valid) nbit = (col_next[(unsigned)a >> 5] >> (a & 31)) & 1u;

    // ---- fold the summaries: zeros before the tile, zeros in the column, last 0 / 1 before the tile ----
    int sumBefore = 0, total = 0, l0 = 0, l1 = 0;
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        total += sv[q].x;
        if (jn < w) { sumBefore += sv[q].x; if (WITH_D) { l0 = max(l0, sv[q].y); l1 = max(l1, sv[q].z); } }
    }
    sumBefore = wave_sum(sumBefore); total = wave_sum(total);
    if (WITH_D) { l0 = wave_max(l0); l1 = wave_max(l1); }
    if (lane == 0) { s_red[wv][0] = sumBefore; s_red[wv][1] = total; s_red[wv][2] = l0; s_red[wv][3] = l1; }
    lds_barrier();
    sumBefore = 0; total = 0; l0 = 0; l1 = 0;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) {
        sumBefore += s_red[q][0]; total += s_red[q][1];
        l0 = max(l0, s_red[q][2]); l1 = max(l1, s_red[q][3]);
    }
    const int Zw = sumBefore;                              // zeros before this tile
    const int C = total;                                   // zeros in the whole column (u->c)
    int m0 = 0, m1 = 0, pd0 = 0, pd1 = 0;
    if (WITH_D) {
        // carry_b = max d over [l_b, S): whole-tile maxima + one partial-tile read (<= 256 positions)
        const int tl0 = l0 ? (l0 - 1) / T : -1, tl1 = l1 ? (l1 - 1) / T : -1;
        const int hi0 = l0 ? min((tl0 + 1) * T, S) : 0, hi1 = l1 ? min((tl1 + 1) * T, S) : 0;
        if (l0 + t < hi0) pd0 = g.d_in[l0 + t];            // the one dependent load; consumed after the scan
        if (l1 + t < hi1) pd1 = g.d_in[l1 + t];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int jn = t + q * BLOCK;
            if (jn < w) { if (jn > tl0) m0 = max(m0, sv[q].w); if (jn > tl1) m1 = max(m1, sv[q].w); }
        }
    }
    PBWT_STAMP(1);

    // ---- the position's own tuple, block scan ----
    Tup me = Tup{0, 0, 0, 0, 0};
    if (valid) { if (y) { me.c1 = 1; me.t0 = d; } else { me.c0 = 1; me.t1 = d; } me.all = d; }
    Tup tot;
    const Tup pre = block_scan_tup<WITH_D>(me, s_tup, tot);
    PBWT_STAMP(2);
    int dn = 0;
    if (WITH_D) {
        m0 = wave_max(max(m0, pd0)); m1 = wave_max(max(m1, pd1));
        if (lane == 0) { s_red[wv][4] = m0; s_red[wv][5] = m1; }
        lds_barrier();
        m0 = 0; m1 = 0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { m0 = max(m0, s_red[q][4]); m1 = max(m1, s_red[q][5]); }
        const int carry0 = l0 ? m0 : k + 1;                // nothing before: p starts at k+1 (pbwtCore.c:489)
        const int carry1 = l1 ? m1 : k + 1;
        const int pin = y ? (pre.c1 ? pre.t1 : max(carry1, pre.all)) : (pre.c0 ? pre.t0 : max(carry0, pre.all));
        dn = max(pin, d);
    }
    PBWT_STAMP(3);
    // ---- stage in LDS in destination order (coalesced stores drain faster at kernel end), then
    //      write out + summaries of site k+1 ----
    const int cw = tot.c0, nvalid = tot.c0 + tot.c1;
    if (valid) {
        const int ldst = y ? cw + pre.c1 : pre.c0;
        s_a[ldst] = a | (int)(nbit << 31);
        if (WITH_D) s_d[ldst] = dn;
    }
    lds_barrier();
    const int oneBase = C + (S - Zw);                      // every earlier tile is full
    const int tz = Zw / T, to = oneBase / T;
    const bool ovalid = FULL || (t < nvalid);
    const bool one = t >= cw;
    const int P = one ? oneBase + (t - cw) : Zw + t;
    const int slot = ovalid ? (one ? 2 + (P / T - to) : (P / T - tz)) : -1;
    unsigned tag = 0;
    if (ovalid) {
        int ao = s_a[t];
        if (SORTED) { if (has_next) tag = (col_next[(unsigned)P >> 5] >> (P & 31)) & 1u; ao |= (int)(tag << 31); }
        else tag = (unsigned)ao >> 31;
        g.a_out[P] = ao;
        if (WITH_D) { dn = s_d[t]; if (P == 0) dn = k + 2; g.d_out[P] = dn; }      // sentinel (pbwtCore.c:507)
    }
    PBWT_STAMP(4);
    if (has_next) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned long long mk = __ballot(slot == s);
            if (mk) {                                      // wave-uniform
                const unsigned long long ones = __ballot(slot == s && tag);
                const unsigned long long zeros = mk & ~ones;
                // within a stream P grows with the lane: the highest lane of a set holds its last position
                const int pz = zeros ? __builtin_amdgcn_readlane(P, 63 - __clzll(zeros)) + 1 : 0;
                const int po = ones ? __builtin_amdgcn_readlane(P, 63 - __clzll(ones)) + 1 : 0;
                int md = 0;
                if (WITH_D) md = wave_max((slot == s) ? dn : 0);
                if (lane == 0) {                           // aggregate in LDS: 16 global atomics per tile, not per wave
                    if (zeros) atomicAdd(&s_acc[s][0], __popcll(zeros));
                    if (WITH_D) {
                        if (pz) atomicMax(&s_acc[s][1], pz);
                        if (po) atomicMax(&s_acc[s][2], po);
                        if (md) atomicMax(&s_acc[s][3], md);
                    }
                }
            }
        }
    }
    if (WITH_D && w == W - 1 && t == 0) g.d_out[M] = k + 2;
    if (t == 0) sm_zero[w] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(5);
    if (has_next) {
        lds_barrier();
        if (t < 16) {
            const int s = t >> 2, f = t & 3;
            const int dt = (s < 2 ? tz : to) + (s & 1);
            const int v = s_acc[s][f];
            if (v && dt < W) {
                int *so = reinterpret_cast<int *>(sm_out + dt) + f;
                if (f == 0) atomicAdd(so, v); else atomicMax(so, v);
            }
        }
    }
    PBWT_STAMP(6);
}

template <bool WITH_D, bool SORTED, int SPT>
__global__ __launch_bounds__(BLOCK) void step1_kernel(StepArgs g) {
    __shared__ Tup s_tup[WAVES];
    __shared__ int s_red[WAVES][6];
    __shared__ int s_acc[4][4];
    __shared__ int s_a[BLOCK];
    __shared__ int s_d[WITH_D ? BLOCK : 1];
    if ((int)(blockIdx.x + 1) * BLOCK <= g.M) step1_body<WITH_D, SORTED, true, SPT>(g, s_a, s_d, s_tup, s_red, s_acc);
    else step1_body<WITH_D, SORTED, false, SPT>(g, s_a, s_d, s_tup, s_red, s_acc);
}

// One site of pbwtCursorForwardsA / ForwardsAD (pbwtCore.c:458-470 / 485-508) for one tile of
// T = 256*E consecutive positions.  grid = W tiles.
template <int E, bool WITH_D, bool SORTED>
__global__ __launch_bounds__(BLOCK) void step_kernel(StepArgs g) {
    constexpr int T = BLOCK * E;
    __shared__ int s_a[T];
    __shared__ int s_d[WITH_D ? T : 1];
    __shared__ Tup s_tup[WAVES];
    __shared__ int s_red[WAVES][6];
    __shared__ int s_acc[4][4];
    if ((int)(blockIdx.x + 1) * T <= g.M) step_body<E, WITH_D, SORTED, true>(g, s_a, s_d, s_tup, s_red, s_acc);
    else step_body<E, WITH_D, SORTED, false>(g, s_a, s_d, s_tup, s_red, s_acc);
}

// ---------------------------------------------------------------------------------------------
// prepare: first site of a pass.  Tags a[i] with y_k[i] from column k and builds that site's tile
// summaries from scratch (plain stores), zeroing the accumulation target of the first step.
struct PrepArgs {
    int *a; const int *d; const uint32_t *col; int4 *summ;
    int k, M, W, wpad, T, sorted, with_d, has_col;
};

__global__ __launch_bounds__(BLOCK) void prepare_kernel(PrepArgs g) {
    __shared__ int s_red[WAVES][4];
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, S = w * g.T;
    int c0 = 0, l0 = 0, l1 = 0, md = 0;
    for (int l = t; l < g.T; l += BLOCK) {
        const int i = S + l;
        if (i < g.M) {
            const int a = g.a[i] & AMASK;
            unsigned y = 0;
            if (g.has_col) {
                const unsigned idx = g.sorted ? (unsigned)i : (unsigned)a;
                y = (g.col[idx >> 5] >> (idx & 31)) & 1u;
            }
            g.a[i] = a | (int)(y << 31);
            if (y == 0) { ++c0; l0 = max(l0, i + 1); } else l1 = max(l1, i + 1);
            if (g.with_d) md = max(md, g.d[i]);
        }
    }
    c0 = wave_sum(c0); l0 = wave_max(l0); l1 = wave_max(l1); md = wave_max(md);
    if (lane == 0) { s_red[wv][0] = c0; s_red[wv][1] = l0; s_red[wv][2] = l1; s_red[wv][3] = md; }
    __syncthreads();
    if (t == 0) {
        c0 = 0; l0 = 0; l1 = 0; md = 0;
        for (int q = 0; q < WAVES; ++q) { c0 += s_red[q][0]; l0 = max(l0, s_red[q][1]); l1 = max(l1, s_red[q][2]); md = max(md, s_red[q][3]); }
        g.summ[w] = make_int4(c0, l0, l1, md);               // batch-relative: step 0 reads buffer 0
        g.summ[(size_t)g.wpad + w] = make_int4(0, 0, 0, 0);
    }
}

// cursor init (pbwtNakedCursorCreate, pbwtCore.c:402-418): a = identity unless given; d = 0 with
// sentinels d[0] = d[M] = k0+1
__global__ void init_state_kernel(int *a, int *d, int M, int Mpad, int k0, int identity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Mpad) {
        if (identity) a[i] = (i < M) ? i : 0;
        else if (i >= M) a[i] = 0;
    }
    if (i <= Mpad) { if (d) d[i] = (i == 0 || i == M) ? k0 + 1 : 0; }
}

// ---------------------------------------------------------------------------------------------
// synthetic panel generator (SURVEY.md §8d recipe in integer arithmetic; the test checker restates it)
__device__ __forceinline__ uint64_t h2(uint64_t seed, uint64_t a, uint64_t b) {
    return sm64(sm64(seed ^ (a * 0xD1B54A32D192ED03ULL)) + b);
}

__global__ __launch_bounds__(BLOCK) void synth_kernel(uint32_t *bits, int M, int k0, int ncols, int wpc,
                                                     uint64_t seed, int kind) {
    __shared__ uint64_t s_fw;
    const int col = blockIdx.y;
    const uint64_t k = (uint64_t)(k0 + col);
    if (kind == 0) {
        // founder word for this site: bit f = founder f carries the derived allele
        if (threadIdx.x < 64) {
            const uint64_t hk = h2(seed ^ 0xB, k, 0);
            const uint32_t e = (uint32_t)(hk & 0xff) % 11u;
            const uint32_t bse = 1u << (31 - e);
            const uint32_t thr = bse / 2 + (uint32_t)((hk >> 8) % (bse / 2));
            const bool on = (uint32_t)(h2(seed ^ 0xA, (uint64_t)threadIdx.x, k) >> 32) < thr;
            const unsigned long long m = __ballot(on);
            if (threadIdx.x == 0) s_fw = m;
        }
        __syncthreads();
    }
    const uint64_t fw = (kind == 0) ? s_fw : 0;
    for (int wd = blockIdx.x * BLOCK + threadIdx.x; wd < wpc; wd += gridDim.x * BLOCK) {
        uint32_t out = 0;
        for (int b = 0; b < 32; ++b) {
            const uint64_t h = (uint64_t)wd * 32 + b;
            if (h >= (uint64_t)M) break;
            uint32_t al;
            if (kind == 1) al = (uint32_t)(h2(seed ^ 0xE, h, k) >> 63);
            else {
                const uint64_t off = h2(seed ^ 0xD, h, 0) % 2048u;
                const uint64_t seg = (k + off) / 2048u;
                const uint32_t F = (uint32_t)(h2(seed ^ 0xC, h, seg) & 63);
                const uint32_t mut = ((uint32_t)(h2(seed ^ 0xE, h, k) >> 32) < 4294967u) ? 1u : 0u;
                al = ((uint32_t)(fw >> F) & 1u) ^ mut;
            }
            out |= al << b;
        }
        bits[(size_t)col * wpc + wd] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// per-site checksums over ring slots: csum[site] += sum_i sm64(i<<32 | v[i]); grid (tiles, sites)
__global__ __launch_bounds__(BLOCK) void checksum_kernel(const int *A, const int *D, size_t strideA, size_t strideD,
                                                        int M, int with_d, unsigned long long *ca,
                                                        unsigned long long *cd, unsigned long long *cy, int y_valid_sites, int packed = 0) {
    __shared__ unsigned long long s_red[WAVES][3];
    const int site = blockIdx.y;
    const int *a = A + (size_t)site * strideA;
    const int *d = D + (size_t)site * strideD;
    unsigned long long sa = 0, sd = 0, sy = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i <= M; i += gridDim.x * BLOCK) {
        if (packed) {                                       // slots hold d | y << 31 and no ids (PBWTAMD_PACKED_CHECKSUM: the packed fill checked position by position)
            const int v = d[i];
            if (i < M) sy += sm64(((uint64_t)i << 32) | ((site < y_valid_sites) ? ((uint32_t)v >> 31) : 0u));
            sd += sm64(((uint64_t)i << 32) | (uint32_t)(i < M ? (v & 0x7fffffff) : v));
            continue;
        }
        if (i < M) {
            const int v = a[i];
            sa += sm64(((uint64_t)i << 32) | (uint32_t)(v & AMASK));
            const uint32_t y = (site < y_valid_sites) ? ((uint32_t)v >> 31) : 0u;
            sy += sm64(((uint64_t)i << 32) | y);
        }
        if (with_d) sd += sm64(((uint64_t)i << 32) | (uint32_t)d[i]);
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sd += __shfl_xor(sd, o); sy += __shfl_xor(sy, o); }
    if (lane_id() == 0) { s_red[wave_id()][0] = sa; s_red[wave_id()][1] = sd; s_red[wave_id()][2] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sa = sd = sy = 0;
        for (int q = 0; q < WAVES; ++q) { sa += s_red[q][0]; sd += s_red[q][1]; sy += s_red[q][2]; }
        atomicAdd(ca + site, sa);
        if (with_d) atomicAdd(cd + site, sd);
        atomicAdd(cy + site, sy);
    }
}

// alleles back to original haplotype order (pbwtWriteHaplotypes, pbwtIO.c:845: hap[a[j]] = y[j]);
// grid (tiles, sites); out[site][hap] = 0/1
__global__ __launch_bounds__(BLOCK) void unsort_alleles_kernel(const int *A, size_t strideA, int M, unsigned char *out) {
    const int s = blockIdx.y;
    const int *a = A + (size_t)s * strideA;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < M; j += gridDim.x * BLOCK) {
        const int v = a[j];
        out[(size_t)s * M + (v & AMASK)] = (unsigned char)((unsigned)v >> 31);
    }
}
__global__ void tags_to_bytes_kernel(const int *a, unsigned char *out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = (unsigned char)((unsigned)a[i] >> 31);
}

// strip tags: out[i] = a[i] & AMASK
__global__ void untag_kernel(const int *a, int *out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = a[i] & AMASK;
}

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
};
// Same-address global atomics serialise chip-wide (~12 ns each): a panel whose matches all have similar lengths (iid: every
// report lands in ~30 bins) would spend seconds there.  So the short lengths are counted in LDS per workgroup first and
// flushed to one of HIST_REP replicas of the low bins; long lengths (spread over many bins) go straight to hist.
constexpr int HIST_LBINS = 2048, HIST_REP = 32;
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
template <bool PACKED>
__device__ __forceinline__ bool coop_walk4(const int *a, const int *d, int from, int dir, int thr, unsigned yi, int M) {
    const int lane = lane_id();
    for (;; from += dir * 256) {
        int wd[4], wy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = from + dir * (lane + 64 * j);
            const int di = (dir < 0) ? p + 1 : p;           // the divergence tested for candidate p (see coop_walk)
            const bool inb = (di >= 0) && (di <= M);
            wd[j] = inb ? __builtin_nontemporal_load(d + di) : 0x7fffffff;
            wy[j] = (p >= 0 && p < M) ? (PACKED ? ((dir < 0) ? __builtin_nontemporal_load(d + p) : wd[j]) : __builtin_nontemporal_load(a + p)) : (int)((yi ^ 1u) << 31);
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
template <bool PACKED>
__global__ __launch_bounds__(BLOCK) void sweep_hist_kernel(SweepArgs g) {
    constexpr int CH = 4;
    __shared__ unsigned s_hist[HIST_LBINS];
    const int site = blockIdx.y, k = g.kbase + site;
    const bool fin = (site == g.final_site);
    const int *a = g.A + (size_t)site * g.strideA;
    const int *d = g.D + (size_t)site * g.strideD;
    const int M = g.M, lane = lane_id();
    for (int x = threadIdx.x; x < HIST_LBINS; x += BLOCK) s_hist[x] = 0;
    __syncthreads();
    // branch-free loads: every address is clamped into [0, M] (index M holds the sentinel d[M]); words of positions beyond M
    // are never used as anything but a right neighbour of an invalid position
    auto WD = [&](int x) -> int {
        const int xc = min(max(x, 0), M);
        return PACKED ? __builtin_nontemporal_load(d + xc) : (__builtin_nontemporal_load(d + xc) | (__builtin_nontemporal_load(a + min(xc, M - 1)) & (int)0x80000000));
    };
    // the words of a group (own 4 chunks + the two halo words, wave-uniform addresses) are requested one iteration ahead of their use
    int nw_[CH], nhl = 0, nhr = 0;
    auto request = [&](int it) {
        const int wb = ((blockIdx.x * g.iters + it) * WAVES + wave_id()) * (64 * CH);
#pragma unroll
        for (int c = 0; c < CH; ++c) nw_[c] = WD(wb + 64 * c + lane);
        nhl = WD(wb - 1);
        nhr = WD(wb + 64 * CH);
    };
    request(0);
    for (int it = 0; it < g.iters; ++it) {
    const int wv = (blockIdx.x * g.iters + it) * WAVES + wave_id();
    const int wbase = wv * (64 * CH);
    int w[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) w[c] = nw_[c];
    const int hl = nhl, hr = nhr;
    request(it + 1);
    if (wbase > M) break;
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
        if (g.ycols) {                                      // this site's sorted bit column (what pack3 encodes)
            const unsigned long long mk = __ballot(valid && yI[c]);
            const int wd = wv * CH + c;
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
                if (!decided) decided = coop_walk4<PACKED>(a, d, wbase - 1, -1, thr, b, M) ? 1 : 2;
                if (decided == 1 && lane == src) { rep[c] = false; pendDn[c] = false; }
            }
        }
    }
    if (mPendDn) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            for (unsigned long long pend = __ballot(pendDn[c]); pend; pend &= pend - 1) {
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
                if (!decided) decided = coop_walk4<PACKED>(a, d, wbase + 64 * CH, +1, thr, b, M) ? 1 : 2;
                if (decided == 1 && lane == src) rep[c] = false;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (rep[c]) {
            const int len = k - min(dI[c], dN[c]);          // (d[i] < d[i+1]) ? k - d[i] : k - d[i+1]   (pbwtMatch.c:131)
            if (len < 0 || len >= g.histlen) atomicExch(g.err, 1);
            else if (len < HIST_LBINS) atomicAdd(&s_hist[len], 1u);
            else atomicAdd(g.hist + len, 1ULL);
        }
    }
    }
    __syncthreads();
    unsigned long long *rep = g.hist_rep + (size_t)((blockIdx.x + 7 * blockIdx.y) % HIST_REP) * HIST_LBINS;
    for (int x = threadIdx.x; x < HIST_LBINS; x += BLOCK) { const unsigned v = s_hist[x]; if (v) atomicAdd(rep + x, (unsigned long long)v); }
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

// single-block exclusive scan of n 64-bit values (in place), total to *total
__global__ __launch_bounds__(1024) void scan_u64_kernel(unsigned long long *v, size_t n, unsigned long long *total,
                                                       unsigned long long base_in) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = base_in;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t b = 0; b < n; b += 1024) {
        const size_t i = b + threadIdx.x;
        const unsigned long long x = (i < n) ? v[i] : 0ULL;
        unsigned long long inc = x;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) pre += s_w[q]; tot += s_w[q]; }
        if (i < n) v[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = s_carry;
}

// large arrays: per-block sums (SCAN_CHUNK values per workgroup) -> scan_u64_kernel over the block sums -> local
// exclusive scan + block offset.  (the single-block kernel streams at one workgroup's bandwidth: 5 ms for 10 M values)
constexpr int SCAN_CHUNK = 4096;
__global__ __launch_bounds__(BLOCK) void scan_u64_blocksum_kernel(const unsigned long long *v, size_t n, unsigned long long *bsum) {
    __shared__ unsigned long long s_w[WAVES];
    const size_t b0 = (size_t)blockIdx.x * SCAN_CHUNK;
    unsigned long long acc = 0;
    for (int x = threadIdx.x; x < SCAN_CHUNK; x += BLOCK) { const size_t i = b0 + x; if (i < n) acc += v[i]; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane_id() == 0) s_w[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int q = 0; q < WAVES; ++q) t += s_w[q]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(BLOCK) void scan_u64_apply_kernel(unsigned long long *v, size_t n, const unsigned long long *boff) {
    __shared__ unsigned long long s_w[WAVES];
    __shared__ unsigned long long s_carry;
    const size_t b0 = (size_t)blockIdx.x * SCAN_CHUNK;
    const int lane = lane_id(), wv = wave_id();
    if (threadIdx.x == 0) s_carry = boff[blockIdx.x];
    __syncthreads();
    for (int x0 = 0; x0 < SCAN_CHUNK; x0 += BLOCK) {
        const size_t i = b0 + x0 + threadIdx.x;
        const unsigned long long x = (i < n) ? v[i] : 0ULL;
        unsigned long long inc = x;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wv) pre += s_w[q]; tot += s_w[q]; }
        if (i < n) v[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// sorted bit columns out of the ring tags: ycol[site][word] (one wave per 64 positions)
__global__ __launch_bounds__(BLOCK) void tags_to_bits_kernel(const int *A, size_t strideA, int M, unsigned long long *ycols,
                                                            int wpc64) {
    const int site = blockIdx.y;
    const int *a = A + (size_t)site * strideA;
    const int nw = (M + 63) / 64;
    for (int wd = blockIdx.x * WAVES + wave_id(); wd < wpc64; wd += gridDim.x * WAVES) {
        const int i = wd * 64 + lane_id();
        const bool one = (wd < nw) && (i < M) && (a[i] < 0);
        const unsigned long long mk = __ballot(one);
        if (lane_id() == 0) ycols[(size_t)site * wpc64 + wd] = mk;
    }
}

// ---------------------------------------------------------------------------------------------
// pack3 encode (pbwtCore.c:240-267) of sorted bit columns.  One block per column; each thread
// owns 64-position words; a run is emitted by the word in which it ENDS.
// bytes for a run of length n (pack3Add, pbwtCore.c:240-252)
__device__ __forceinline__ int p3_nbytes(int n) {
    int c = 0;
    if (n >= 63488) { c = n / 63488; n -= c * 63488; }       // rare: keep the division off the common path
    if (n >= 2048) { ++c; n &= 0x7ff; }
    if (n >= 64) { ++c; n &= 0x3f; }
    if (n) ++c;
    return c;
}
__device__ __forceinline__ uint8_t *p3_emit(uint8_t *o, unsigned v, int n) {
    const uint8_t top = (uint8_t)(v << 7);
    while (n >= 63488) { *o++ = top | 0x7f; n -= 63488; }
    if (n >= 2048) { *o++ = top | 0x60 | (uint8_t)(n >> 11); n &= 0x7ff; }
    if (n >= 64) { *o++ = top | 0x40 | (uint8_t)(n >> 6); n &= 0x3f; }
    if (n) *o++ = top | (uint8_t)n;
    return o;
}

// MODE 0: colBytes[col] = encoded size; MODE 1: write bytes at colOffset[col]
// NT threads per column: the loop over chunks of NT words is a chain of barriers and dependent loads (latency of ONE
// workgroup, whatever the batch), so wide columns take 1024 threads.
template <int MODE, int NT = BLOCK>
__global__ __launch_bounds__(NT) void pack3_kernel(const unsigned long long *ycols, int wpc64, int M,
                                                     unsigned long long *colBytes, uint8_t *out) {
    constexpr int NWV = NT / 64;
    __shared__ int s_wi[NWV];
    __shared__ int s_carry_start;       // start position of the run open at the chunk boundary
    __shared__ int s_carry_bytes;       // bytes emitted so far in this column
    const int col = blockIdx.x;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64;
    const int lane = lane_id(), wv = wave_id();
    if (threadIdx.x == 0) { s_carry_start = 0; s_carry_bytes = 0; }
    __syncthreads();
    uint8_t *obase = (MODE == 1) ? out + colBytes[col] : nullptr;
    for (int b = 0; b < nw; b += NT) {
        const int wd = b + threadIdx.x;
        unsigned long long cur = 0, trans = 0;
        int nbits = 0;
        if (wd < nw) {
            cur = y[wd];
            nbits = min(64, M - wd * 64);
            const unsigned long long prevbit = (wd > 0) ? (y[wd - 1] >> 63) : 0ULL;
            trans = cur ^ ((cur << 1) | prevbit);          // bit p set: position starts a new run
            if (wd == 0) trans &= ~1ULL;                   // position 0 opens the first run, closes nothing
            if (nbits < 64) trans &= (1ULL << nbits) - 1ULL;
        }
        // last run start at or before the beginning of this word: max-scan of last transition pos
        int lastT = trans ? (wd * 64 + 63 - __clzll(trans)) : -1;
        int incl = lastT;
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl = max(incl, v); }
        if (lane == 63) s_wi[wv] = incl;
        int exclT = __shfl_up(incl, 1); if (lane == 0) exclT = -1;
        __syncthreads();
        int preT = -1;
        for (int q = 0; q < NWV; ++q) if (q < wv) preT = max(preT, s_wi[q]);
        int chunkLast = -1;
        for (int q = 0; q < NWV; ++q) chunkLast = max(chunkLast, s_wi[q]);
        int open = max(max(exclT, preT), -1);
        if (open < 0) open = s_carry_start;                // run opened in an earlier chunk (or at 0)
        // runs closed by this word: one per transition, plus the final run if this word holds M-1
        const bool lastWord = (wd == nw - 1);
        int myBytes = 0;
        {
            unsigned long long tr = trans; int st = open;
            while (tr) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; tr &= tr - 1; myBytes += p3_nbytes(pz - st); st = pz; }
            if (lastWord) myBytes += p3_nbytes(M - st);
        }
        // exclusive scan of myBytes within the chunk
        int inc = myBytes;
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        __syncthreads();
        if (lane == 63) s_wi[wv] = inc;
        __syncthreads();
        int preB = s_carry_bytes, totB = 0;
        for (int q = 0; q < NWV; ++q) { if (q < wv) preB += s_wi[q]; totB += s_wi[q]; }
        if (MODE == 1 && myBytes) {
            uint8_t *o = obase + preB + inc - myBytes;
            unsigned long long tr = trans; int st = open;
            while (tr) {
                const int pz = wd * 64 + __ffsll((long long)tr) - 1; tr &= tr - 1;
                // value of the run [st,pz) = bit at st
                const unsigned v = (unsigned)((y[st >> 6] >> (st & 63)) & 1ULL);
                o = p3_emit(o, v, pz - st); st = pz;
            }
            if (lastWord) { const unsigned v = (unsigned)((y[st >> 6] >> (st & 63)) & 1ULL); o = p3_emit(o, v, M - st); }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry_bytes += totB; if (chunkLast >= 0) s_carry_start = chunkLast; }
        __syncthreads();
    }
    if (MODE == 0 && threadIdx.x == 0) colBytes[col] = (unsigned long long)s_carry_bytes;
}

// pack3 encode, wave-regional: a wave owns a contiguous region of 64*IT words of the column and walks it 64 words
// (= one coalesced 512-byte load) at a time; all IT loads are issued up front.  Inside the wave the start of the run open
// at a word is an exclusive max-scan over the lanes (DPP) carried across the iterations — no barrier; across the waves of
// the column ONE LDS exchange of (first / last transition, bytes) fixes the run open at each region's start and the byte
// bases.  A run is emitted by the word in which it ends; its value is the last bit of the previous word and alternates from
// there.  (pack3_kernel above does the same with a barrier chain per 1024-word chunk: 16 chunks x 5 barriers at M = 1 M.)
template <int MODE, int NT, int IT>
__global__ __launch_bounds__(NT) void pack3v2_kernel(const unsigned long long *ycols, int wpc64, int M,
                                                       unsigned long long *colBytes, uint8_t *out) {
    constexpr int NWV = NT / 64;
    __shared__ int s_last[NWV], s_first[NWV], s_inner[NWV];
    const int col = blockIdx.x, lane = lane_id(), wv = wave_id();
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = wv * 64 * IT;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    const int hi0 = (base > 0 && base <= nw) ? (int)(y[base - 1] >> 32) : 0;      // the word before the region (its last bit matters)
    // transitions of word i of this lane: bit p set = position 64 wd + p starts a new run.  prevHi carries the previous
    // iteration's last word across the loop.
    auto transitions = [&](int i, int &prevHi) -> unsigned long long {
        const int wd = base + i * 64 + lane;
        const int hi = (int)(cur[i] >> 32);
        const int ph = lane_shr1(hi, prevHi);
        prevHi = __builtin_amdgcn_readlane(hi, 63);
        unsigned long long tr = cur[i] ^ ((cur[i] << 1) | (unsigned long long)((unsigned)ph >> 31));
        if (wd == 0) tr &= ~1ULL;                            // position 0 opens the first run, closes nothing
        const int nbits = M - wd * 64;
        if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
        return tr;
    };
    // ---- pass A: first / last transition of the region, bytes of the runs that start at a transition of the region and end in it
    int carryT = -1, firstT = -1, inner = 0, prevHi = hi0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        unsigned long long tr = transitions(i, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1, tf = tr ? wd * 64 + __ffsll((long long)tr) - 1 : -1;
        const int inc = wave_iscan_max(tl + 1);             // 1 + last transition up to and including this lane (0 = none)
        int st = max(lane_shr1(inc, 0) - 1, carryT);        // start of the run open at this word; -1 = it began before the region
        int bytes = 0;
        for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; if (st >= 0) bytes += p3_nbytes(pz - st); st = pz; }
        inner += wave_sum(bytes);
        const unsigned long long has = __ballot(tf >= 0);
        if (has) {
            if (firstT < 0) firstT = __builtin_amdgcn_readlane(tf, __ffsll((long long)has) - 1);
            carryT = __builtin_amdgcn_readlane(inc, 63) - 1;
        }
    }
    if (lane == 0) { s_last[wv] = carryT; s_first[wv] = firstT; s_inner[wv] = inner; }
    __syncthreads();
    // ---- the waves before this one: run open at the region's start, byte base
    int openW = 0, baseB = 0, total = 0;
    {
        int open = 0;                                        // start of the run open at wave q's region (position 0 opens the first run)
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            const int lq = s_last[q], fq = s_first[q];
            const bool ownsLast = (q * 64 * IT < nw) && ((q + 1) * 64 * IT >= nw);
            const int wb = (fq >= 0 ? p3_nbytes(fq - open) : 0) + s_inner[q] + (ownsLast ? p3_nbytes(M - (lq >= 0 ? lq : open)) : 0);
            if (q == wv) { openW = open; baseB = total; }
            total += wb;
            if (lq >= 0) open = lq;
        }
    }
    if (MODE == 0) { if (threadIdx.x == 0) colBytes[col] = (unsigned long long)total; return; }
    // ---- pass B: emit.  The run open at the region's start now has a known start (openW).
    // Scattered single-byte stores to HBM are slow (partial-sector writes): a column of up to P3_STAGE bytes — all but iid-like
    // columns of wide panels — is assembled in LDS and copied out with consecutive lanes writing consecutive bytes.
    constexpr int P3_STAGE = 32768;
    __shared__ uint8_t s_stage[MODE == 1 ? P3_STAGE : 1];
    const bool staged = total <= P3_STAGE;
    uint8_t *obase = (staged ? s_stage : out + colBytes[col]) + baseB;
    // (a rolled loop over freshly reloaded, L2-hot words: keeping all IT words live through the emission code spills registers)
    carryT = openW; prevHi = hi0;
    int done = 0;                                            // bytes emitted so far by this wave
    unsigned long long nxt = (base + lane < nw) ? y[base + lane] : 0ULL;
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        if (base + i * 64 >= nw) break;
        const unsigned long long cw = nxt;
        nxt = (wd + 64 < nw && i + 1 < IT) ? y[wd + 64] : 0ULL;
        const int hiPrevIter = prevHi;                       // the value of the run open at this word = the last bit before it
        unsigned long long tr;
        {
            const int hi = (int)(cw >> 32);
            const int ph0 = lane_shr1(hi, prevHi);
            prevHi = __builtin_amdgcn_readlane(hi, 63);
            tr = cw ^ ((cw << 1) | (unsigned long long)((unsigned)ph0 >> 31));
            if (wd == 0) tr &= ~1ULL;
            const int nbits = M - wd * 64;
            if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
        }
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1;
        const int inc = wave_iscan_max(tl + 1);
        int st = max(lane_shr1(inc, 0) - 1, carryT);
        const bool lastWord = (wd == nw - 1);
        int bytes = 0;
        { int s2 = st; for (unsigned long long t2 = tr; t2; t2 &= t2 - 1) { const int pz = wd * 64 + __ffsll((long long)t2) - 1; bytes += p3_nbytes(pz - s2); s2 = pz; } if (lastWord) bytes += p3_nbytes(M - s2); }
        const int incB = wave_iscan_sum(bytes);
        const int ph = lane_shr1((int)(cw >> 32), hiPrevIter);   // cross-lane: outside the divergent branch below
        if (bytes) {
            uint8_t *o = obase + done + incB - bytes;
            unsigned v = (wd == 0) ? (unsigned)(cw & 1ULL) : ((unsigned)ph >> 31);
            for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; o = p3_emit(o, v, pz - st); st = pz; v ^= 1u; }
            if (lastWord) p3_emit(o, v, M - st);
        }
        done += __builtin_amdgcn_readlane(incB, 63);
        const int wl = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (wl >= 0) carryT = wl;
    }
    if (staged) {
        __syncthreads();
        uint8_t *dst = out + colBytes[col];
        for (int x = threadIdx.x; x < total; x += NT) dst[x] = s_stage[x];
    }
}

// pack3 encode, region-parallel (three launches per batch of columns): a column is cut into REGIONS of 64*IT words, one wave
// each, and no wave waits for another.
//   p3r_scan_kernel    per region: first / last transition, bytes of the runs that start and end inside it
//   p3r_combine_kernel per column (one wave): the run open at each region's start (max-scan over the regions' last transitions),
//                      the regions' byte bases (sum-scan), the column's size
//   p3r_emit_kernel    per region: emission at the column's offset + the region's base
// pack3v2_kernel does the same inside one workgroup per column; its 16 waves x 16 serial iterations at M = 1 M are a latency
// chain (0.27 ms per 512 columns) where this form runs 245 single-iteration waves per column.
struct P3Region { int firstT, lastT, inner, pad; };           // after combine: {openW, baseB, -, -}

template <int IT>
__device__ __forceinline__ unsigned long long p3r_transitions(const unsigned long long (&cur)[IT], int i, int base, int lane, int M, int &prevHi) {
    const int wd = base + i * 64 + lane;
    const int hi = (int)(cur[i] >> 32);
    const int ph = lane_shr1(hi, prevHi);                    // previous word's high half (lane 0: the last word before this iteration)
    prevHi = __builtin_amdgcn_readlane(hi, 63);
    unsigned long long tr = cur[i] ^ ((cur[i] << 1) | (unsigned long long)((unsigned)ph >> 31));
    if (wd == 0) tr &= ~1ULL;                                // position 0 opens the first run, closes nothing
    const int nbits = M - wd * 64;
    if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
    return tr;
}

template <int IT>
__global__ __launch_bounds__(BLOCK) void p3r_scan_kernel(const unsigned long long *ycols, int wpc64, int M, int R, P3Region *regs) {
    const int col = blockIdx.y, reg = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (reg >= R) return;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = reg * 64 * IT;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    int prevHi = (base > 0 && base <= nw) ? (int)(y[base - 1] >> 32) : 0;
    int carryT = -1, firstT = -1, inner = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        unsigned long long tr = p3r_transitions<IT>(cur, i, base, lane, M, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1, tf = tr ? wd * 64 + __ffsll((long long)tr) - 1 : -1;
        const int inc = wave_iscan_max(tl + 1);             // 1 + last transition up to and including this lane (0 = none)
        int st = max(lane_shr1(inc, 0) - 1, carryT);        // start of the run open at this word; -1 = it began before the region
        int bytes = 0;
        for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; if (st >= 0) bytes += p3_nbytes(pz - st); st = pz; }
        inner += wave_sum(bytes);
        const unsigned long long has = __ballot(tf >= 0);
        if (has) {
            if (firstT < 0) firstT = __builtin_amdgcn_readlane(tf, __ffsll((long long)has) - 1);
            carryT = __builtin_amdgcn_readlane(inc, 63) - 1;
        }
    }
    if (lane == 0) regs[(size_t)col * R + reg] = P3Region{firstT, carryT, inner, 0};
}

// one wave per column: lanes = regions, 64 at a time with carries
__global__ __launch_bounds__(BLOCK) void p3r_combine_kernel(int M, int R, int words_per_region, int ncols, P3Region *regs, unsigned long long *colBytes) {
    const int col = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (col >= ncols) return;
    P3Region *rg = regs + (size_t)col * R;
    const int nw = (M + 63) / 64;
    int openCarry = 0, byteCarry = 0;                        // position 0 opens the first run
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        const P3Region v = (r < R) ? rg[r] : P3Region{-1, -1, 0, 0};
        const int inc = wave_iscan_max(v.lastT + 1);
        const int prevLast = lane_shr1(inc, 0) - 1;          // last transition in the earlier regions of this group of 64, -1 = none
        const int open = (prevLast >= 0) ? prevLast : openCarry;
        const bool ownsLast = (r < R) && (r * words_per_region < nw) && ((r + 1) * words_per_region >= nw);
        const int wb = (r < R) ? (v.firstT >= 0 ? p3_nbytes(v.firstT - open) : 0) + v.inner + (ownsLast ? p3_nbytes(M - (v.lastT >= 0 ? v.lastT : open)) : 0) : 0;
        const int incB = wave_iscan_sum(wb);
        if (r < R) rg[r] = P3Region{open, byteCarry + incB - wb, 0, 0};
        const int lastAll = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (lastAll >= 0) openCarry = lastAll;
        byteCarry += __builtin_amdgcn_readlane(incB, 63);
    }
    if (lane == 0) colBytes[col] = (unsigned long long)byteCarry;
}

template <int IT>
__global__ __launch_bounds__(BLOCK) void p3r_emit_kernel(const unsigned long long *ycols, int wpc64, int M, int R, const P3Region *regs,
                                                           const unsigned long long *colOff, uint8_t *out) {
    const int col = blockIdx.y, reg = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (reg >= R) return;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = reg * 64 * IT;
    if (base >= nw) return;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    const P3Region rg = regs[(size_t)col * R + reg];
    uint8_t *obase = out + colOff[col] + rg.lastT;           // .lastT holds the region's byte base after the combine
    int prevHi = (base > 0) ? (int)(y[base - 1] >> 32) : 0;
    int carryT = rg.firstT;                                  // .firstT holds the start of the run open at the region's start
    int done = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        const int hiPrevIter = prevHi;
        unsigned long long tr = p3r_transitions<IT>(cur, i, base, lane, M, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1;
        const int inc = wave_iscan_max(tl + 1);
        int st = max(lane_shr1(inc, 0) - 1, carryT);
        const bool lastWord = (wd == nw - 1);
        int bytes = 0;
        { int s2 = st; for (unsigned long long t2 = tr; t2; t2 &= t2 - 1) { const int pz = wd * 64 + __ffsll((long long)t2) - 1; bytes += p3_nbytes(pz - s2); s2 = pz; } if (lastWord) bytes += p3_nbytes(M - s2); }
        const int incB = wave_iscan_sum(bytes);
        const int ph = lane_shr1((int)(cur[i] >> 32), hiPrevIter);   // cross-lane: outside the divergent branch below
        if (bytes) {
            uint8_t *o = obase + done + incB - bytes;
            unsigned v = (wd == 0) ? (unsigned)(cur[i] & 1ULL) : ((unsigned)ph >> 31);   // value of the run open at this word = the last bit before it
            for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; o = p3_emit(o, v, pz - st); st = pz; v ^= 1u; }
            if (lastWord) p3_emit(o, v, M - st);
        }
        done += __builtin_amdgcn_readlane(incB, 63);
        const int wl = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (wl >= 0) carryT = wl;
    }
}

// ---------------------------------------------------------------------------------------------
// pack3 decode (unpack3, pbwtCore.c:279-305).
__device__ __forceinline__ int p3_len(uint8_t b) {
    b &= 0x7f;
    return b < 64 ? b : (b < 96 ? (b - 64) << 6 : (b - 96) << 11);
}
// pass 1: per block of DEC_CHUNK bytes, total run length
constexpr int DEC_CHUNK = 4096;
__global__ __launch_bounds__(BLOCK) void dec_sum_kernel(const uint8_t *z, size_t nz, unsigned long long *blockSum) {
    __shared__ unsigned long long s_w[WAVES];
    const size_t b0 = (size_t)blockIdx.x * DEC_CHUNK;
    unsigned long long s = 0;
    for (int q = threadIdx.x; q < DEC_CHUNK; q += BLOCK) { const size_t i = b0 + q; if (i < nz) s += (unsigned)p3_len(z[i]); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane_id() == 0) s_w[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) { s = 0; for (int q = 0; q < WAVES; ++q) s += s_w[q]; blockSum[blockIdx.x] = s; }
}
// pass 2: with exclusive block offsets: colStart[c] = byte index of the first byte of column c
// (start position divisible by M and non-empty run); colStart[N] = nz written by the host.
__global__ __launch_bounds__(BLOCK) void dec_colstart_kernel(const uint8_t *z, size_t nz, const unsigned long long *blockOff,
                                                            int M, long long N, long long *colStart) {
    __shared__ unsigned long long s_w[WAVES];
    __shared__ unsigned long long s_carry;
    const size_t b0 = (size_t)blockIdx.x * DEC_CHUNK;
    if (threadIdx.x == 0) s_carry = blockOff[blockIdx.x];
    __syncthreads();
    for (int q0 = 0; q0 < DEC_CHUNK; q0 += BLOCK) {
        const size_t i = b0 + q0 + threadIdx.x;
        const unsigned len = (i < nz) ? (unsigned)p3_len(z[i]) : 0u;
        unsigned long long inc = len;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
        const unsigned long long start = pre + inc - len;
        if (i < nz && len && start % (unsigned long long)M == 0) {
            const unsigned long long c = start / (unsigned long long)M;
            if ((long long)c < N) colStart[c] = (long long)i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}
// pass 2b: a well-formed panel has every column start found (a run never straddles a column boundary,
// pbwtCore.c:254-267), strictly increasing, at most M bytes per column.  Checked BEFORE any expand: a crafted file
// otherwise leaves colStart[c] = -1 (the memset) and the expand would index z[] and y[] out of bounds.
__global__ void dec_validate_kernel(const long long *colStart, long long N, long long nz, int M, int *err) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const long long bs = colStart[c], be = (c + 1 < N) ? colStart[c + 1] : nz;
    if (bs < 0 || be <= bs || be > nz || be - bs > (long long)M) atomicExch(err, 2);
}

// pass 3: expand columns [c0, c0+nc) into sorted bit columns (zero-initialised by the caller).
// One block per column; runs of ones set bits.  Malformed input (already rejected by dec_validate_kernel on the
// upload path) cannot write outside the column: bounds are re-checked and the accumulators are 64-bit.
__global__ __launch_bounds__(BLOCK) void dec_expand_kernel(const uint8_t *z, const long long *colStart, long long c0, int M,
                                                          unsigned long long *ycols, int wpc64, int *err) {
    __shared__ long long s_w[WAVES];
    __shared__ long long s_carry;
    const long long c = c0 + blockIdx.x;
    const long long bs = colStart[c], be = colStart[c + 1];
    unsigned long long *y = ycols + (size_t)blockIdx.x * wpc64;
    if (bs < 0 || be < bs || be - bs > (long long)M) { if (threadIdx.x == 0) atomicExch(err, 2); return; }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long b = bs; b < be; b += BLOCK) {
        const long long i = b + threadIdx.x;
        const uint8_t byte = (i < be) ? z[i] : 0;
        const int len = (i < be) ? p3_len(byte) : 0;
        long long inc = len;
        for (int o = 1; o < 64; o <<= 1) { long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
        const long long start = pre + inc - len;
        if (start + len > (long long)M) atomicExch(err, 2);
        if (len && (byte & 0x80) && start < (long long)M) {
            int lo = (int)start, hi = (int)min(start + len, (long long)M);      // [lo,hi)
            while (lo < hi) {
                const int wd = lo >> 6, bo = lo & 63;
                const int take = min(64 - bo, hi - lo);
                const unsigned long long mk = (take == 64) ? ~0ULL : (((1ULL << take) - 1ULL) << bo);
                if (take == 64) y[wd] = mk; else atomicOr(&y[wd], mk);
                lo += take;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_carry != (long long)M) atomicExch(err, 3);
}

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
